"""VAE at the reference's own image size (1024x1024 -> 128x128 latents): runs, is finite, agrees with a tiled
evaluation of the same images at batch 1 (micro-batch independence), and how long it takes."""
import sys, time, torch
sys.path.insert(0, '.')
from diffuman4d_amd.host.vae import AutoencoderKL, VAEConfig
from diffuman4d_amd.host.weights import vae_param_shapes, random_state_dict
cfg = VAEConfig()
vae = AutoencoderKL(cfg, random_state_dict(vae_param_shapes(cfg), 1, "cuda"), "cuda")
H = W = 1024
n = 9  # > micro batch (7): exercises the split
img = (torch.rand(n, 3, H, W, device="cuda") * 2 - 1)
noise = torch.randn(n, 4, H // 8, W // 8, device="cuda")
print("micro batch at 1024^2:", vae.micro_batch(H, W))
torch.cuda.synchronize(); t0 = time.perf_counter()
z = vae.encode_scaled(img, noise)
torch.cuda.synchronize(); te = time.perf_counter() - t0
t0 = time.perf_counter()
out = vae.decode_to_images(z)
torch.cuda.synchronize(); td = time.perf_counter() - t0
print(f"encode {n} imgs {te*1e3:.0f} ms ({n*4.9/te:.0f} TF/s), decode {td*1e3:.0f} ms ({n*10.5/td:.0f} TF/s), finite {bool(torch.isfinite(out.float()).all())}, out {tuple(out.shape)}")
z1 = vae.encode_scaled(img[8:9], noise[8:9])
o1 = vae.decode_to_images(z1)
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
print(f"same image in a different micro-batch (other tile configs => other summation order): latents rel_l2 {rel(z1, z[8:9]):.2e}, "
      f"images rel_l2 {rel(o1, out[8:9]):.2e}; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GB")
