"""Sample rocm-smi while the engine runs a long GEMM loop (random vs zero operands)."""
import sys, os, subprocess, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lvt_amd.hip import gemm as G
dev = "cuda:0"
M = 8192
def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--showperflevel"], capture_output=True, text=True).stdout
    keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("Power (W)", "sclk", "mclk", "fclk", "Max Graphics Package Power", "Performance Level"))]
    return " | ".join(k.split("GPU[0]")[-1].strip(" :\t") for k in keep if "GPU[0]" in k)
print("idle:", smi(), flush=True)
for name, fill in [("random", None), ("zeros", 0.0)]:
    xb = torch.randn(M, 4096, device=dev); wb = torch.randn(8192, 4096, device=dev); ob = torch.empty(M, 8192, device=dev)
    if fill is not None: xb.fill_(fill); wb.fill_(fill)
    stop = False
    samples = []
    def sampler():
        time.sleep(1.0)
        while not stop:
            samples.append(smi()); time.sleep(1.0)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    while time.time() - t0 < 5.0:
        for _ in range(20): G.gemm(xb, wb, ob, M, 8192, 4096)
        n += 20; torch.cuda.synchronize()
    b.record(); torch.cuda.synchronize()
    stop = True; th.join()
    ms = a.elapsed_time(b) / n
    print("%s: %.1f TF" % (name, 2 * M * 8192 * 4096 / ms / 1e9))
    for s_ in samples[:4]: print("   ", s_)
