"""PR-DVQVAE2 train-step time (bench.py leg) under the environment's switches (A/B runs on one box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
leg = bench.VqvaeLeg("cuda:0", 1, 0, 0, 32, 4)
for i in range(4): leg.step(i)
torch.cuda.synchronize(); t = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for i in range(n): leg.step(4 + i)
torch.cuda.synchronize()
print("vqvae step %.3f ms  presplit_off=%s" % ((time.time() - t) * 1e3 / n, os.environ.get("LVT_NO_PRESPLIT")))
