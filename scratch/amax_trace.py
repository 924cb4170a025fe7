"""which call sites still need a stand-alone max |.| pass (LVT_AMAX_TRACE=1): python scratch/amax_trace.py vqvae|dsfvt"""
import os, sys
os.environ["LVT_AMAX_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from lvt_amd.hip import binding as L
which = sys.argv[1]
leg = bench.VqvaeLeg("cuda:0", 1, 0, 0, 32, 2) if which == "vqvae" else bench.DsfvtLeg("cuda:0", 1, 0, 0, 64, 2)
for i in range(2): leg.step(i)
L.AMAX_TRACE.clear(); n0 = L.AMAX_FALLBACKS[0]
leg.step(2)
torch.cuda.synchronize()
print(which, "fallback passes per step:", L.AMAX_FALLBACKS[0] - n0)
for k, v in sorted(L.AMAX_TRACE.items(), key=lambda kv: -kv[1]): print("  %d x %s" % (v, k))
