"""median ms of one bench.py leg over N steps (same box A/B of env switches): python tools/profile/leg_time.py dsfvt|vqvae [steps]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
which = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = "cuda:0"; torch.cuda.set_device(0)
leg = bench.VqvaeLeg(dev, 1, 0, 0, 32, 4) if which == "vqvae" else bench.DsfvtLeg(dev, 1, 0, 0, 64, 4)
for i in range(5): leg.step(i)
torch.cuda.synchronize()
ts = []
for i in range(steps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); leg.step(5 + i); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
# the bench times its steps FREE-RUNNING (one synchronisation around K steps); the per-step synchronised median above it starts every
# step with an empty queue and so carries the host's start-of-step latency (~1.7 ms for DSFVT) and every change of host work 1:1
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps): leg.step(5 + steps + i)
torch.cuda.synchronize(); free = (time.perf_counter() - t0) / steps * 1e3
print(which, "free-running %.3f ms per step |" % free, {k: v for k, v in os.environ.items() if k.startswith("LVT_")}, "median %.3f ms  p10 %.3f  p90 %.3f" % (statistics.median(ts), sorted(ts)[len(ts) // 10], sorted(ts)[-len(ts) // 10 - 1]))
