#!/usr/bin/env python3
"""Diagnostic (CPU, oracle only): which fp16 storage roundings of the HIP plan contribute how much to the one-forward error?
The fp32 oracle is run with its activations rounded to fp16 at the points where the HIP plan stores a tensor, by class:
  carrier  - the residual stream: conv_in, resnet outputs, transformer-block residual adds, proj_out(+residual), up/down convs
  interior - everything stored inside a block: norm outputs, conv1 output, proj_in, qkv / q, attention outputs, GEGLU output
Usage: python tools/diag/fp16_noise_budget.py [tiny]"""
import os, sys, time
import torch
import torch.nn as nn
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import unet_ref as U

R = lambda t: t.half().float()
MODE = {"carrier": False, "interior": False, "tf32carrier": False}   # tf32carrier: transformer-internal residual adds kept in fp32
rc = lambda t: R(t) if MODE["carrier"] else t
ri = lambda t: R(t) if MODE["interior"] else t


def block_forward(self, hidden_states, encoder_hidden_states=None):
    a1 = self.attn1
    n = ri(self.norm1(hidden_states))
    B, N, C = n.shape
    h = a1.heads
    q, k, v = ri(a1.to_q(n)), ri(a1.to_k(n)), ri(a1.to_v(n))
    sp = lambda t: t.view(B, -1, h, C // h).transpose(1, 2)
    o = ri(F.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, N, C))
    rt = (lambda t: t) if MODE["tf32carrier"] else rc
    hidden_states = rt(a1.to_out[0](o) + hidden_states)
    n2 = ri(self.norm2(hidden_states))
    x = self.attn2(n2, encoder_hidden_states=encoder_hidden_states)       # interior roundings of attn2 are tiny (0.45 % of flops): skipped
    hidden_states = rt(x + hidden_states)
    n3 = ri(self.norm3(hidden_states))
    g = ri(self.ff.net[0](n3))
    hidden_states = rc(self.ff.net[2](g) + hidden_states)
    return hidden_states


def t2d_forward(self, hidden_states, encoder_hidden_states=None):
    b, _, h, w = hidden_states.shape
    residual = hidden_states
    hidden_states = (lambda t: t if MODE["tf32carrier"] else rc(t))(self.proj_in(ri(self.norm(hidden_states))))
    inner = hidden_states.shape[1]
    hidden_states = hidden_states.permute(0, 2, 3, 1).reshape(b, h * w, inner)
    for blk in self.transformer_blocks:
        hidden_states = blk(hidden_states, encoder_hidden_states=encoder_hidden_states)
    hidden_states = hidden_states.reshape(b, h, w, inner).permute(0, 3, 1, 2).contiguous()
    return rc(self.proj_out(hidden_states) + residual)


def res_forward(self, x, temb):
    h = self.conv1(ri(self.nonlinearity(self.norm1(x))))
    h = ri(h + self.time_emb_proj(self.nonlinearity(temb))[:, :, None, None])
    h = self.conv2(ri(self.nonlinearity(self.norm2(h))))
    if self.conv_shortcut is not None:
        x = ri(self.conv_shortcut(x))
    return rc(x + h)


U.BasicTransformerBlockRef.forward = block_forward
U.Transformer2DModelRef.forward = t2d_forward
U.ResnetBlock2DRef.forward = res_forward
_down, _up = U.Downsample2DRef.forward, U.Upsample2DRef.forward
U.Downsample2DRef.forward = lambda self, x: rc(_down(self, x))
U.Upsample2DRef.forward = lambda self, x: rc(_up(self, x))

tiny = len(sys.argv) > 1 and sys.argv[1] == "tiny"
torch.manual_seed(0)
ref = U.UNet2DConditionModelRef(**(U.TINY_CONFIG if tiny else {})).eval()
U.set_visual_cross_attention_adapter_ref(ref, (5,))
ref.conv_in.register_forward_hook(lambda m, i, o: rc(o))
with torch.no_grad():
    for p in ref.parameters():
        p.copy_(R(p))                     # fp16-representable weights, as the HIP plan packs them
g = torch.Generator().manual_seed(3)
S = 16 if tiny else 64
x, text, ip = torch.randn(1, 4, S, S, generator=g), R(torch.randn(1, 77, 768, generator=g)), R(torch.randn(1, 1, 768, generator=g))
outs = {}
for name, (c, i, tf) in {"fp32": (False, False, False), "carrier only": (True, False, False), "interior only": (False, True, False),
                         "both (= HIP plan)": (True, True, False), "both, fp32 hs0..hs2 inside transformer blocks": (True, True, True)}.items():
    MODE["carrier"], MODE["interior"], MODE["tf32carrier"] = c, i, tf
    t0 = time.time()
    with torch.no_grad():
        outs[name] = ref(x, torch.tensor(481), encoder_hidden_states=(text, ip)).sample.double()
    e = ((outs[name] - outs["fp32"]).norm() / outs["fp32"].norm()).item()
    print(f"{name:22s} rel-L2 vs fp32 = {e:.3e}   ({time.time() - t0:.1f} s)", flush=True)
