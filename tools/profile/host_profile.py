"""cProfile of the HOST side of DSFVT train steps (where do the ~23 ms of Python per step go): python tools/profile/host_profile.py [steps]"""
import os, sys, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
dev = "cuda:0"; torch.cuda.set_device(0)
leg = bench.DsfvtLeg(dev, 1, 0, 0, 64, 4)
for i in range(4): leg.step(i)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
# the backward pass runs on the autograd engine's own thread: profile it there too
import threading
prof_bw = cProfile.Profile()
state = {"on": False}
def hook(*a):            # a full-backward pre-hook is awkward here: use threading.setprofile for new threads instead
    pass
pr = cProfile.Profile()
threading.setprofile(lambda *a: None)
pr.enable()
for i in range(n): leg.step(4 + i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO(); ps = pstats.Stats(pr, stream=s).sort_stats("tottime"); ps.print_stats(28)
print(s.getvalue()[:6000])
