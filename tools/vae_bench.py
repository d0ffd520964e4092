#!/usr/bin/env python3
"""VAE decode / encode timing: bs=16, 64x64 latents <-> 512x512 images, random-init SD-v1.5 VAE."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photoverse_amd.vae import AutoencoderKL
torch.manual_seed(0)
vae = AutoencoderKL().to("cuda")
z = torch.randn(16, 4, 64, 64, device="cuda")
for _ in range(2):
    vae.decode(z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = vae.decode(z).sample
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
flop = 2.51e12 * 16      # 2.51 TFLOP per 512x512 image (counted from the decoder's layer shapes; conv-dominated)
print(f"VAE decode bs=16 512x512: {dt * 1e3:.1f} ms  ({16 / dt:.1f} images/s, ~{flop / dt / 1e12:.0f} TFLOP/s), finite={bool(torch.isfinite(out).all())}")

x = torch.rand(16, 3, 512, 512, device="cuda") * 2 - 1
for _ in range(2):
    vae.encode(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    dist = vae.encode(x).latent_dist
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
rec = next(p.rec for k, p in vae._plans.items() if k[0] == "enc")
flop = sum(t[1] for t in rec.tags)     # GEMM / conv launches of one sub-batch pass
print(f"VAE encode bs=16 512x512: {dt * 1e3:.1f} ms  ({16 / dt:.1f} images/s, {flop / 1e12:.2f} TFLOP per pass -> ~{flop / dt / 1e12:.0f} TFLOP/s), "
      f"finite={bool(torch.isfinite(dist.mean).all())}")
