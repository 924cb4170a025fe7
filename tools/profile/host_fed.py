"""The bench step fed from host memory through DevicePrefetcher (bench.host_fed): python tools/profile/host_fed.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
torch.cuda.set_device(0)
vq = bench.VqvaeLeg("cuda:0", 1, 0, 0, 32, 4)
ds = bench.DsfvtLeg("cuda:0", 1, 0, 0, 64, 4)
for i in range(3):
    vq.step(2 * i); vq.step(2 * i + 1); ds.step(i)
torch.cuda.synchronize()
print(bench.host_fed(vq, ds, int(sys.argv[1]) if len(sys.argv) > 1 else 8, 1, "cuda:0"))
