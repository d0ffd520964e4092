#!/usr/bin/env python3
"""Full-size parity run (long: ~10 min of host CPU): SD-v1.5-shaped UNet (859.5 M params, random init, shared weights),
B=1, 64x64 latents, P=1, guidance 7.5, DPM-Solver++ T-step CFG loop on the HIP path vs the fp32 CPU oracle.
Prints the latent rel-L2 after selected steps.  Usage (GPU box): python3 tools/full_parity.py [steps=50]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.infer_ref import denoise_ref, draw_noise_ref
from oracle.unet_ref import UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
from photoverse_amd.pipeline import DenoiseLoop
from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
torch.manual_seed(0)
ref = UNet2DConditionModelRef().eval()
set_visual_cross_attention_adapter_ref(ref, (5,))
hip = UNet2DConditionModel()
set_visual_cross_attention_adapter(hip, (5,))
hip.load_state_dict(ref.state_dict())
hip.to("cuda")
g = torch.Generator().manual_seed(31)
B, P = 1, 1
cond = (torch.randn(B, 77, 768, generator=g), torch.randn(B, P, 768, generator=g))
uncond = (torch.randn(B, 77, 768, generator=g), torch.randn(B, P, 768, generator=g))
noise = draw_noise_ref(B, 4, 64, seed=6)
loop = DenoiseLoop(hip, B, 64, P, T, 7.5)
loop.set_conditioning(tuple(t.cuda() for t in cond), tuple(t.cuda() for t in uncond))
loop.reset(noise)
got = []
for _ in range(T):
    loop.step()
    got.append(loop.latents.detach().cpu().clone())
t0 = time.time()
exp = []
denoise_ref(ref, noise, cond, uncond, guidance_scale=7.5, timesteps=T, collect=exp)
print(f"oracle: {T} steps in {time.time() - t0:.0f} s on {torch.get_num_threads()} threads")
for i in sorted(set([0, 4, 9, 24, T // 2, T - 1])):
    if i < T:
        a, b = got[i].double(), exp[i].double()
        print(f"step {i + 1:3d}: latents rel-L2 = {((a - b).norm() / b.norm()).item():.3e}   max|diff| = {(a - b).abs().max().item():.3e}   |latents| rms = {b.pow(2).mean().sqrt().item():.3f}")
