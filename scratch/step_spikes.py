import sys, os, time, gc
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
def run(tag, n=30):
    st0 = torch.cuda.memory_stats()
    t0 = time.perf_counter(); enq = []
    for i in range(n):
        a = time.perf_counter(); bench.vqvae_step(model, opts, data, i); enq.append(time.perf_counter() - a)
    torch.cuda.synchronize(); tot = time.perf_counter() - t0
    st1 = torch.cuda.memory_stats()
    print("%s: %.2f ms/step; enqueue ms:" % (tag, tot / n * 1e3), " ".join("%.0f" % (e * 1e3) for e in enq))
    print("   device_alloc +%d device_free +%d retries +%d reserved %.1f GB allocated peak %.1f GB" % (
        st1["num_device_alloc"] - st0["num_device_alloc"], st1["num_device_free"] - st0["num_device_free"],
        st1["num_alloc_retries"] - st0["num_alloc_retries"], st1["reserved_bytes.all.current"] / 2**30, st1["allocated_bytes.all.peak"] / 2**30))
run("default")
gc.collect(); gc.disable()
run("gc disabled")
gc.enable()
gc.freeze()
run("gc frozen")
