"""One leg of bench.py alone (for rocprofv3 passes): python tools/profile/bench_leg.py vqvae|dsfvt|combined [steps] [warmup]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
which = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = "cuda:0"
torch.cuda.set_device(0)
legs = []
if which in ("vqvae", "combined"):
    legs.append(bench.VqvaeLeg(dev, 1, 0, 0, 32, 4))
if which in ("dsfvt", "combined"):
    legs.append(bench.DsfvtLeg(dev, 1, 0, 0, 64, 4))


def step(i):
    for leg in legs:
        for j in range(2 if (which == "combined" and isinstance(leg, bench.VqvaeLeg)) else 1):
            leg.step(2 * i + j)


for i in range(warmup):
    step(i)
torch.cuda.synchronize()
for i in range(steps):
    step(warmup + i)
torch.cuda.synchronize()
# engine wrapper calls per step, counted the way bench.py counts them (lvt_amd.hip.binding.KernelTimer keys): what the
# bench line compares with when it quotes a PMC file of this run (traffic_stale)
from lvt_amd.hip import binding as L
L.TIMER = L.KernelTimer()
step(warmup + steps)
torch.cuda.synchronize()
calls = sum(1 for k, _, _, _ in L.TIMER.records if k.startswith(("conv_", "gemm_", "attn_")))
L.TIMER = None
print("done", which, steps, warmup, "ENGINE_CALLS_PER_STEP", calls)
