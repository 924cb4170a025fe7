"""aten ops that launch kernels in one train step of a leg, with their Python call sites: python scratch/aten_ops.py dsfvt|vqvae"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from torch.profiler import profile, ProfilerActivity
which = sys.argv[1]
leg = bench.VqvaeLeg("cuda:0", 1, 0, 0, 32, 2) if which == "vqvae" else bench.DsfvtLeg("cuda:0", 1, 0, 0, 64, 2)
for i in range(3): leg.step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    leg.step(3)
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::cat", "aten::clone", "aten::contiguous", "aten::sum", "aten::index", "aten::masked_fill", "aten::div")
rows = {}
for ev in prof.events():
    if ev.name in want:
        st = [s for s in (ev.stack or []) if "lvt_amd" in s or "bench.py" in s][:2]
        key = (ev.name, str(ev.input_shapes)[:60], " < ".join(s.split("/")[-1][:70] for s in st))
        rows[key] = rows.get(key, 0) + 1
names = {}
for ev in prof.events():
    if ev.name.startswith("aten::"): names[ev.name] = names.get(ev.name, 0) + 1
print("all aten ops of the step (every thread):", sorted(names.items(), key=lambda kv: -kv[1])[:25])
for k, v in sorted(rows.items(), key=lambda kv: -kv[1])[:40]:
    print("%3d x %-16s %-60s %s" % (v, k[0], k[1], k[2]))
