"""What happens between two DSFVT train steps: host time of the per-pass preparation (_begin_pass: max |.| records, weight images)
and whether the GPU is idle while it runs.  python tools/profile/step_boundary.py [steps]"""
import os, sys, time, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
dev = "cuda:0"; torch.cuda.set_device(0)
leg = bench.DsfvtLeg(dev, 1, 0, 0, 64, 4)
model = leg.model
st = torch.cuda.current_stream()
orig = model._begin_pass
rec = {"host": [], "drained_before": [], "drained_after": [], "gap": []}
last_end = [None]
def wrapped():
    d0 = st.query()
    ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
    t0 = time.perf_counter()
    orig()
    t1 = time.perf_counter()
    ev1 = torch.cuda.Event(enable_timing=True); ev1.record()
    rec.setdefault("span", []).append((ev0, ev1))
    d1 = st.query()
    rec["host"].append((t1 - t0) * 1e3); rec["drained_before"].append(d0); rec["drained_after"].append(d1)
    rec.setdefault("ev", []).append((last_end[0], ev0))
model._begin_pass = wrapped
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for i in range(steps + 4):
    leg.step(i)
    e = torch.cuda.Event(enable_timing=True); e.record(); last_end[0] = e
torch.cuda.synchronize()
gaps = [a.elapsed_time(b) * 1e3 for a, b in rec["ev"][4:] if a is not None]
print("mode", os.environ.get("LVT_P2", "(default)"))
print("_begin_pass host time: median %.2f ms" % statistics.median(rec["host"][4:]))
print("stream already drained when _begin_pass starts: %d of %d; when it ends: %d of %d" % (sum(rec["drained_before"][4:]), steps, sum(rec["drained_after"][4:]), steps))
spans = [a.elapsed_time(b) * 1e3 for a, b in rec["span"][4:]]
print("GPU time from the start to the end of _begin_pass's launches (their kernels take ~0.15 ms): median %.1f us -- the rest is the GPU waiting for the host" % statistics.median(spans))
print("GPU time between the end of a step's last kernel and the event recorded at the start of the next _begin_pass: median %.1f us" % statistics.median(gaps))
