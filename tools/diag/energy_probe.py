#!/usr/bin/env python3
"""Is a launch's time the SUM of its resource floors because the package power limit binds?  Two probes, sustained timing (0.3 s of launches first):
(1) the fused attn2 kernel (C = 320, B = 16, N = 4096, P = 1) and the row-owning LayerNorm + GEGLU projection on random and on all-zero operands - the
    instruction stream is identical, only the bits toggling in the data paths differ;
(2) an MFMA-bound launch (3x3 conv 320 -> 320 @ 64 x 64, half-chip form: 128 tiles) and an HBM-bound launch (GroupNorm-apply + SiLU of the same tensor), each
    alone and both side by side on two streams: t(both) against max(t_conv, t_gn) (perfect overlap) and t_conv + t_gn (none).
usage (GPU box): python tools/diag/energy_probe.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.ops import Recorder, ACT_SILU  # noqa: E402
dev = torch.device("cuda")
torch.manual_seed(0)


def sustained(fn, warm=400, reps=200):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


B, n, H, d, NT, P = 16, 4096, 8, 40, 77, 1
C = H * d
for name, mk in (("randn", lambda *s: torch.randn(*s, device=dev).half()), ("zeros", lambda *s: torch.zeros(*s, device=dev).half())):
    hs, kvt, kvip = mk(B * n, C), mk(B * NT, 2 * C), mk(B * P, 2 * C)
    wq, wo = mk(C, C) * C ** -0.5, mk(C, C) * C ** -0.5
    rec = Recorder(dev)
    vn = torch.zeros(B, H, P, device=dev)
    kimg, vimg = rec.xattn_pack_kv(kvt[:, :C], kvt[:, C:], kvip[:, :C], kvip[:, C:], batch=B, heads=H, d=d, nt=NT, nip=P, vnorm=vn)
    rec.run()
    torch.cuda.synchronize()
    r2 = Recorder(dev)
    r2.cross_attention_fused(hs, wq, r2.pack_wo_for_fused(wo), torch.zeros(C, device=dev), kimg, vimg, batch=B, nq=n, heads=H, d=d, nt=NT, nip=P,
                             ln_gamma=torch.ones(C, device=dev), ln_beta=torch.zeros(C, device=dev), w_text=1.0, w_ip=1.0)
    t_x = sustained(r2.run)
    r3 = Recorder(dev)
    w1 = mk(8 * C, C) * C ** -0.5
    r3.row_gemm(hs, w1, bias=torch.zeros(8 * C, device=dev), ln_gamma=torch.ones(C, device=dev), ln_beta=torch.zeros(C, device=dev), geglu=True)
    t_g = sustained(r3.run)
    print(f"{name:6s}  fused attn2 C=320: {t_x:6.1f} us    LayerNorm + GEGLU projection (row-owning): {t_g:6.1f} us", flush=True)

# (2) overlap of an MFMA-bound and an HBM-bound launch
x = torch.randn(B * n, C, device=dev).half()
w = (torch.randn(C, 9 * C, device=dev) * (9 * C) ** -0.5).half()
conv = Recorder(dev)
conv.big_min = 128
geo = dict(batch=B // 2, hin=64, win=64, hout=64, wout=64)
conv.gemm(x[: B // 2 * n], w, conv=geo, colstats=True)
gn = Recorder(dev)
y = torch.randn(B * n, C, device=dev).half()
gn.groupnorm(y, torch.ones(C, device=dev), torch.zeros(C, device=dev), batch=B, hw=n, act=ACT_SILU)
print("launches:", conv.tags[0][0], conv.tags[0][3], "|", [t[0] if t else None for t in gn.tags])
t_c, t_n = sustained(conv.run), sustained(gn.run)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    conv.run(s1.cuda_stream)
    gn.run(s2.cuda_stream)


torch.cuda.synchronize()
for _ in range(300):
    both()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
for _ in range(200):
    both()
torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
e1.record()
torch.cuda.synchronize()
t_b = e0.elapsed_time(e1) / 200 * 1e3
print(f"conv (128 tiles) alone {t_c:.1f} us, GroupNorm + SiLU alone {t_n:.1f} us, side by side {t_b:.1f} us per pair   (max {max(t_c, t_n):.1f}, sum {t_c + t_n:.1f})")
