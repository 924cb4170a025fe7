"""Event-timed engine launches of one DSFVT train step in issue order (no profiler): python tools/profile/step_events.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from lvt_amd.hip import binding as L
dev = "cuda:0"; torch.cuda.set_device(0)
leg = bench.DsfvtLeg(dev, 1, 0, 0, 64, 4)
for i in range(4): leg.step(i)
torch.cuda.synchronize()
L.TIMER = L.KernelTimer()
leg.step(4)
torch.cuda.synchronize()
recs = L.TIMER.records; L.TIMER = None
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for key, flops, a, b in recs[:n]:
    us = a.elapsed_time(b) * 1e3
    print("%-16s %8.1f us  %7.2f GF  %6.1f TF/s" % (key, us, flops / 1e9, flops / us / 1e6))
