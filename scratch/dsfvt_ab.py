"""DSFVT train-step time (bench.py leg) with the attention path given by the environment (A/B runs on the same box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
leg = bench.DsfvtLeg("cuda:0", 1, 0, 0, 64, 4)
for i in range(4): leg.step(i)
torch.cuda.synchronize(); t = time.time()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for i in range(n): leg.step(4 + i)
torch.cuda.synchronize()
print("dsfvt step %.2f ms  max mem %.2f GB  flash_off=%s" % ((time.time() - t) * 1e3 / n, torch.cuda.max_memory_allocated() / 2**30, os.environ.get("LVT_NO_FLASH_ATTENTION")))
