import sys, time, torch
sys.path.insert(0, '.')
from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
from diffuman4d_amd.host.weights import vae_param_shapes, random_state_dict
cfg = VAEConfig()
vae = AutoencoderKL(cfg, random_state_dict(vae_param_shapes(cfg), 1, "cuda"), "cuda")
H, W = 576, 320
img = (torch.rand(8, 3, H, W, device="cuda") * 2 - 1)
noise = torch.randn(8, 4, H // 8, W // 8, device="cuda")
def t(f, n=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
z = vae.encode_scaled(img, noise)
te = t(lambda: vae.encode_scaled(img, noise))
td = t(lambda: vae.decode_to_images(z))
print(f"VAE 576x320: encode 8 imgs {te*1e3:.1f} ms ({8*0.78/te:.0f} TF/s), decode 8 imgs {td*1e3:.1f} ms ({8*1.76/td:.0f} TF/s)")
print("finite:", bool(torch.isfinite(vae.decode_to_images(z).float()).all()))
