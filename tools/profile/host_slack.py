"""How much host slack does a DSFVT train step have?  A busy-wait of X us is added to the per-pass preparation (_begin_pass) and,
separately, to the start of the backward pass; a step that is GPU-bound there does not get slower.  python tools/profile/host_slack.py"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
dev = "cuda:0"; torch.cuda.set_device(0)
leg = bench.DsfvtLeg(dev, 1, 0, 0, 64, 4)
model = leg.model
orig = model._begin_pass
delay = [0.0]
def spin(us):
    t = time.perf_counter() + us * 1e-6
    while time.perf_counter() < t: pass
def wrapped():
    orig(); spin(delay[0])
model._begin_pass = wrapped
def run(n=20):
    ts = []
    for i in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); leg.step(i); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return statistics.median(ts)
def run_nosync(n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): leg.step(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for i in range(5): leg.step(i)
for us in (0, 1000, 2000, 4000, 6000, 8000, 12000):
    delay[0] = us
    print("busy-wait %4d us in _begin_pass: step %.3f ms (synchronised per step), %.3f ms (free-running)" % (us, run(), run_nosync()))
