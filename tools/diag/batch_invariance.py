#!/usr/bin/env python3
"""Diagnostic: one full-size UNet forward at bs=16 vs the bs=1 forward of sample i (no CFG, no scheduler)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter

torch.manual_seed(0)
hip = UNet2DConditionModel()
set_visual_cross_attention_adapter(hip, (5,))
hip.to("cuda")
g = torch.Generator().manual_seed(77)
B = 16
x, text, ip = torch.randn(B, 4, 64, 64, generator=g), torch.randn(B, 77, 768, generator=g), torch.randn(B, 1, 768, generator=g)
with torch.no_grad():
    full = hip(x.cuda(), torch.tensor(481), encoder_hidden_states=(text.cuda(), ip.cuda())).sample.cpu()
    for i in (0, 7, 15):
        one = hip(x[i:i + 1].cuda(), torch.tensor(481), encoder_hidden_states=(text[i:i + 1].cuda(), ip[i:i + 1].cuda())).sample.cpu()
        d = (full[i:i + 1].double() - one.double())
        print(f"SPLITK_MAX={os.environ.get('PV_SPLITK_MAX', 'default')} sample {i}: rel-L2 = {(d.norm() / one.double().norm()).item():.3e}  max|d| = {d.abs().max().item():.3e}")
