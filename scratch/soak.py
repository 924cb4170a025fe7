"""Soak: many VQ-VAE and DSFVT train steps on a FIXED synthetic batch (so the loss must fall): finite losses,
flat memory, stable step time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
dev = "cuda:0"
torch.cuda.set_device(0)
cfg, model = bench.build_vqvae(dev, 1234)
optimizers, _ = model.configure_optimizers_and_checkpointers()
rng = np.random.RandomState(0)
# smooth synthetic clips (learnable): low-frequency patterns
base = rng.rand(32, 1, 3, 8, 8).astype(np.float32)
clips = np.repeat(np.repeat(np.repeat(base, 16, 1), 8, 3), 8, 4)
data = [{"image_sequence": clips[i]} for i in range(32)]
log = []
t0 = time.time()
for i in range(1500):
    losses = bench.vqvae_step(model, optimizers, data, i)
    if i % 250 == 0 or i == 1499:
        torch.cuda.synchronize()
        l = {k: float(v) for k, v in losses.items()}
        log.append((i, l, torch.cuda.memory_allocated() >> 20, torch.cuda.max_memory_allocated() >> 20, time.time() - t0))
        print("vqvae step %4d  %s  alloc %d MiB  peak %d MiB  t %.1fs" % (i, l, log[-1][2], log[-1][3], log[-1][4]), flush=True)
assert all(np.isfinite(list(l.values())).all() for _, l, *_ in log)
assert log[-1][1]["loss_reconstruction"] < 0.25 * log[0][1]["loss_reconstruction"], "reconstruction loss did not fall"
assert log[-1][3] == log[2][3], "peak memory grew"
print("VQ-VAE soak OK")
del model, optimizers
torch.cuda.empty_cache()

from lvt_amd.config import get_cfg
from lvt_amd.data.dataset_mapper import prepare_slices_batch
from lvt_amd.modeling import build_model
from lvt_amd.utils.events import EventStorage
cfg = get_cfg(); cfg.merge_from_file(os.path.join(ROOT, "configs/vt/DSFVT.yaml")); cfg.MODEL.DEVICE = dev
torch.manual_seed(7)
model = build_model(cfg); model.train()
optimizers, _ = model.configure_optimizers_and_checkpointers()
v = cfg.MODEL.AUTOREGRESSIVE.VT
g = torch.Generator(device="cpu").manual_seed(4321)
codes = torch.randint(0, v.NV, (16, 16, v.NC, 16, 16), generator=g).to(dev)
abcs = [(int(a), 0, 0) for a in torch.randint(v.N_PRIME, 16, (16,), generator=g)]
ctx, sl, sidx, ign = prepare_slices_batch(codes, abcs, v.STRIDE, v.KERNEL, v.N_PRIME, v.PAD_VALUE)
first = last = None; peak = []
for i in range(300):
    with EventStorage(i):
        loss = model.compute_supervised_loss(ctx, sl, sidx, ign)["loss_cross_entropy"]
    loss.backward(); model.finish_gradient_sync()
    for o in optimizers: o["optimizer"].step()
    for o in optimizers: o["optimizer"].zero_grad()
    if i % 50 == 0 or i == 299:
        lv = float(loss); peak.append(torch.cuda.max_memory_allocated() >> 20)
        first = lv if first is None else first; last = lv
        print("dsfvt step %3d  loss %.4f  peak %d MiB" % (i, lv, peak[-1]), flush=True)
assert np.isfinite(last) and last < first, "DSFVT loss did not fall on a fixed batch"
assert peak[-1] == peak[1], "peak memory grew"
print("DSFVT soak OK")
