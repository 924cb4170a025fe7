import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from lvt_amd.hip import binding as L
cfg, model = bench.build_vqvae("cuda:0", 1)
opts, _ = model.configure_optimizers_and_checkpointers()
clips = torch.rand(32, 16, 3, 64, 64).cuda()
data = [{"image_sequence": clips[i]} for i in range(32)]
for i in range(5): bench.vqvae_step(model, opts, data, i)
torch.cuda.synchronize()
print("mode", L.get_math_mode())
# (a) enqueue time per step (CPU only) and pipelined total
t0 = time.perf_counter(); enq = []
for i in range(20):
    a = time.perf_counter(); bench.vqvae_step(model, opts, data, i); enq.append(time.perf_counter() - a)
torch.cuda.synchronize(); tot = time.perf_counter() - t0
print("pipelined: %.2f ms/step; cpu enqueue per step (ms):" % (tot / 20 * 1e3), " ".join("%.1f" % (e * 1e3) for e in enq))
# (b) synced per step
lat = []
for i in range(10):
    torch.cuda.synchronize(); a = time.perf_counter(); bench.vqvae_step(model, opts, data, i); torch.cuda.synchronize(); lat.append(time.perf_counter() - a)
print("synced per-step latency (ms):", " ".join("%.1f" % (e * 1e3) for e in lat))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5): bench.vqvae_step(model, opts, data, i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
