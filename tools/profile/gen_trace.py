"""Generation profile helper: runs bench_generate with a reduced number of generated frames."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
r = bench.bench_generate("cuda:0", int(sys.argv[1]) if len(sys.argv) > 1 else 64)
print(r)
