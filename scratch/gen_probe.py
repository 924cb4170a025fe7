"""Probe of the generation leg (bench.py:bench_generate) at a bounded size: host RSS, device memory in use and wall time
after every phase, to find what grows with the number of decode steps / videos.  Usage: gen_probe.py BATCH GROUP_ROWS [SLICES]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "golden")]
import psutil  # noqa: E402
import torch  # noqa: E402

B, ROWS = int(sys.argv[1]), int(sys.argv[2])
NPRIME = 16 - int(sys.argv[3]) if len(sys.argv) > 3 else 5
PHASES = sys.argv[4] if len(sys.argv) > 4 else "vq,sample"
proc = psutil.Process()
T0 = time.perf_counter()


def mark(what, sync=True):
    if sync:
        torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    vm = psutil.virtual_memory()
    print("%7.2fs %-28s rss %6.2f GB  host avail %6.1f/%5.1f GB  vram used %6.2f GB (torch reserved %6.2f)"
          % (time.perf_counter() - T0, what, proc.memory_info().rss / 2**30, vm.available / 2**30, vm.total / 2**30,
             (total - free) / 2**30, torch.cuda.memory_reserved() / 2**30), flush=True)


import lvt_amd.modeling.meta_arch.vt as vtmod  # noqa: E402
from lvt_amd.config import get_cfg  # noqa: E402
from lvt_amd.modeling import build_model  # noqa: E402
vtmod.DECODE_GROUP_ROWS = ROWS
cfgs = []
for path in ("configs/vt/DSFVT.yaml", "configs/vqvae/PR-DVQVAE2.yaml"):
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, path))
    cfg.MODEL.DEVICE = "cuda:0"
    cfg.OUTPUT_DIR = "/tmp/lvt_probe"
    cfgs.append(cfg)
cfgs[0].TEST.EVALUATORS = "VTSampler"
torch.manual_seed(1)
vt, vqvae = build_model(cfgs[0]).eval(), build_model(cfgs[1]).eval()
mark("models built")
frames = torch.rand(B, NPRIME, 3, 64, 64).to("cuda:0")
orig = vt.model.encoder.forward_tokens
nslice = [0]


def enc(*a, **k):
    out = orig(*a, **k)
    mark("  slice %d encoder pass issued (no sync)" % nslice[0], sync=False)
    nslice[0] += 1
    return out


vt.model.encoder.forward_tokens = enc
for it in range(2):
    with torch.no_grad():
        out = vqvae([{"image_sequence": frames[i]} for i in range(B)], mode="inference")
        lat = torch.stack([o["latent"] for o in out])
        mark("run %d: vq encode" % it)
        video = lat.new_zeros(B, 16, lat.shape[2], 16, 16)
        video[:, :NPRIME] = lat
        if "sample" in PHASES:
            sample = vt.sample_video(video.transpose(1, 2).contiguous(), n_prime=NPRIME)
        else:
            sample = torch.randint(0, 512, (B, lat.shape[2], 16, 16, 16), device="cuda:0")
        mark("run %d: sampled" % it)
        if "vq" not in PHASES:
            rec = sample
            continue
        rec = vqvae.decode(sample.transpose(1, 2).reshape(B * 16, -1, 16, 16))
        mark("run %d: decoded %d frames" % (it, B * 16))
print("OK", tuple(rec.shape), int(sample.max()))
