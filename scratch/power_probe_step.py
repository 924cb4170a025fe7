"""rocm-smi sampled once per second during the real VQ-VAE train step loop and the DSFVT loop of bench.py."""
import sys, os, subprocess, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    keep = [l for l in out.splitlines() if "GPU[0]" in l and any(k in l for k in ("Power (W)", "sclk"))]
    return " | ".join(k.split("GPU[0]")[-1].strip(" :\t") for k in keep)
dev = "cuda:0"; torch.cuda.set_device(0)
cfg, model = bench.build_vqvae(dev, 1)
opt, _ = model.configure_optimizers_and_checkpointers()
clips = torch.rand(32, 16, 3, 64, 64).to(dev)
data = [{"image_sequence": clips[i]} for i in range(32)]
stop = False; samples = []
def sampler():
    time.sleep(1.5)
    while not stop:
        samples.append(smi()); time.sleep(1.0)
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(10): bench.vqvae_step(model, opt, data, n); n += 1
    torch.cuda.synchronize()
stop = True; th.join()
print("VQ-VAE train step loop: %.2f ms/step" % ((time.time() - t0) / n * 1e3))
for s_ in samples[:4]: print("   ", s_)
