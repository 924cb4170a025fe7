"""Is the host ahead of the GPU?  At the end of every host phase of a bench leg's step (forward enqueued, backward enqueued, optimizer
step enqueued) ask the stream whether it has already drained (query() == True: the GPU is waiting for the host), and time the host
side of each phase.  python tools/profile/host_ahead.py dsfvt|vqvae [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from lvt_amd.utils.events import EventStorage
which = sys.argv[1] if len(sys.argv) > 1 else "dsfvt"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
torch.cuda.set_device(0)
leg = bench.VqvaeLeg(dev, 1, 0, 0, 32, 4) if which == "vqvae" else bench.DsfvtLeg(dev, 1, 0, 0, 64, 4)
for i in range(3): leg.step(i)
torch.cuda.synchronize()
st = torch.cuda.current_stream()
names = ["forward", "backward", "optimizer.step", "zero_grad"]
host = [0.0] * 4; drained = [0] * 4
t_all = time.perf_counter()
for i in range(steps):
    t0 = time.perf_counter()
    if which == "vqvae":
        with EventStorage(i):
            losses = leg.model(leg.batches[i % leg.nbatches], mode="supervised")
        loss = sum(losses.values())
    else:
        ctx, sl, sidx, ign = leg.batches[i % leg.nbatches]
        with EventStorage(i):
            loss = leg.model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
    t1 = time.perf_counter(); d0 = st.query()
    loss.backward()
    t2 = time.perf_counter(); d1 = st.query()
    for o in leg.optimizers: o["optimizer"].step()
    t3 = time.perf_counter(); d2 = st.query()
    for o in leg.optimizers: o["optimizer"].zero_grad()
    t4 = time.perf_counter(); d3 = st.query()
    for k, (a, b) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4))): host[k] += b - a
    for k, d in enumerate((d0, d1, d2, d3)): drained[k] += int(d)
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / steps * 1e3
print("%s: %.2f ms per step (wall, %d steps); host time per step %.2f ms" % (which, wall, steps, sum(host) / steps * 1e3))
for k in range(4):
    print("  %-15s host %6.2f ms   stream already drained at its end in %d of %d steps" % (names[k], host[k] / steps * 1e3, drained[k], steps))
