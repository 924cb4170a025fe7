import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch, bench
from lvt_amd.data.prefetch import DevicePrefetcher
from lvt_amd.utils.events import EventStorage
torch.cuda.set_device(0)
vq = bench.VqvaeLeg("cuda:0", 1, 0, 0, 32, 4)
ds = bench.DsfvtLeg("cuda:0", 1, 0, 0, 64, 4)
for i in range(3):
    vq.step(2 * i); vq.step(2 * i + 1); ds.step(i)
torch.cuda.synchronize()
vq_host = [[{"image_sequence": c.cpu().numpy()} for c in clips] for clips in vq.clips]
ds_host = []
for ctx, sl, sidx, ign in ds.batches:
    c, s_, i_, g_ = ctx.cpu().numpy(), sl.cpu().numpy(), sidx.cpu().numpy(), ign.cpu().numpy()
    ds_host.append([{"context": c[j], "slice": s_[j], "slice_idx": i_[j], "ignore_mask": g_[j]} for j in range(c.shape[0])])
def T(name, fn, n=8):
    fn(2); torch.cuda.synchronize(); t = time.perf_counter(); fn(n); torch.cuda.synchronize()
    print("%-52s %.2f ms per step" % (name, (time.perf_counter() - t) / n * 1e3))
def stage_only(n):
    for _ in DevicePrefetcher(itertools.islice(itertools.cycle(vq_host), 2 * n), "cuda:0"): pass
    for _ in DevicePrefetcher(itertools.islice(itertools.cycle(ds_host), n), "cuda:0"): pass
T("staging only (2 vq + 1 ds batch)", stage_only)
def opt(leg):
    for o in leg.optimizers: o["optimizer"].step()
    for o in leg.optimizers: o["optimizer"].zero_grad()
def vq_resident(n):
    for i in range(2 * n):
        with EventStorage(i): l = vq.model(vq.batches[i % 4], mode="supervised")
        sum(l.values()).backward(); opt(vq)
T("vq x2, device-resident list[dict]", vq_resident)
def vq_fed(n):
    it = iter(DevicePrefetcher(itertools.islice(itertools.cycle(vq_host), 2 * n), "cuda:0"))
    for i in range(2 * n):
        with EventStorage(i): l = vq.model(next(it), mode="supervised")
        sum(l.values()).backward(); opt(vq)
T("vq x2, host-fed", vq_fed)
def ds_direct(n):
    for i in range(n): ds.step(i)
T("ds, compute_supervised_loss on device tensors", ds_direct)
dev_lists = [[{k: torch.as_tensor(v).cuda() for k, v in d.items()} for d in b] for b in ds_host]
def ds_list(n):
    for i in range(n):
        with EventStorage(i): l = ds.model(dev_lists[i % 4], mode="supervised")["loss_cross_entropy"]
        l.backward(); opt(ds)
T("ds, model(list[dict] of device tensors)", ds_list)
def ds_fed(n):
    it = iter(DevicePrefetcher(itertools.islice(itertools.cycle(ds_host), n), "cuda:0"))
    for i in range(n):
        with EventStorage(i): l = ds.model(next(it), mode="supervised")["loss_cross_entropy"]
        l.backward(); opt(ds)
T("ds, host-fed", ds_fed)
