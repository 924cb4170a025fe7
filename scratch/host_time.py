"""Host enqueue time vs GPU time of one leg, plus a cProfile of the host side and the f16x2 fallback count.
usage: python scratch/host_time.py vqvae|dsfvt [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
from lvt_amd.hip import binding as L

which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = "cuda:0"
torch.cuda.set_device(0)
leg = bench.VqvaeLeg(dev, 1, 0, 0, 32, 4) if which == "vqvae" else bench.DsfvtLeg(dev, 1, 0, 0, 64, 4)
for i in range(3):
    leg.step(i)
torch.cuda.synchronize()
f0 = L.AMAX_FALLBACKS[0]
t0 = time.perf_counter()
for i in range(steps):
    leg.step(3 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s math=%s: host enqueue %.2f ms/step, total %.2f ms/step, amax fallbacks/step %.1f" %
      (which, L.get_math_mode(), (t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3, (L.AMAX_FALLBACKS[0] - f0) / steps))
if L.AMAX_TRACE is not None:
    L.AMAX_TRACE.clear()
    leg.step(50)
    for k, v in sorted(L.AMAX_TRACE.items(), key=lambda kv: -kv[1]):
        print("  fallback x%d: %s" % (v, k))
pr = cProfile.Profile()
pr.enable()
for i in range(4):
    leg.step(100 + i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
