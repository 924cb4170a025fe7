"""The DSFVT leg of bench.py on its own (for the profilers): python scratch/bench_dsfvt_only.py [steps] [warmup]."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.cuda.set_device(0)
out = bench.bench_dsfvt("cuda:0", 1, 0, steps, warm, 64, 4, strict_f32=False, cpu_seconds=0.0)
print(json.dumps({k: v for k, v in out.items() if k != "roofline"}))
