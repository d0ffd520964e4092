#!/usr/bin/env python3
"""Diagnostic: where the fused attn2 kernel's output differs from the four-launch path (row / column pattern of bad elements)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.ops import Recorder

def h16(*s, scale=1.0, seed=0):
    return (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale).half()

B, H, d, NT = 2, 8, 40, 77
C = H * d
for n, p, ln in ((4096, 1, True), (128, 5, False)):
    hs = h16(B * n, C, seed=40); hs[:, ::7] += 1.5
    kvt, kvip = h16(B * NT, 2 * C, seed=41), h16(B * p, 2 * C, seed=42)
    wq, wo = h16(C, C, scale=C ** -0.5, seed=43), h16(C, C, scale=C ** -0.5, seed=44)
    bo = torch.randn(C, generator=torch.Generator().manual_seed(45))
    gamma = 1.0 + 0.2 * torch.randn(C, generator=torch.Generator().manual_seed(46)); beta = 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(47))
    rec = Recorder("cuda")
    dhs, dt, di = hs.cuda(), kvt.cuda(), kvip.cuda()
    kimg, vimg = rec.xattn_pack_kv(dt[:, :C], dt[:, C:], di[:, :C], di[:, C:], batch=B, heads=H, d=d, nt=NT, nip=p)
    out, _ = rec.cross_attention_fused(dhs, wq.cuda(), rec.pack_wo_for_fused(wo.cuda()), bo.cuda(), kimg, vimg, batch=B, nq=n, heads=H, d=d,
                                       nt=NT, nip=p, ln_gamma=gamma.cuda() if ln else None, ln_beta=beta.cuda() if ln else None, w_text=1.0, w_ip=1.0)
    n2 = rec.layernorm(dhs, gamma.cuda(), beta.cuda()) if ln else dhs
    q = rec.gemm(n2, wq.cuda(), rows_per_image=n)
    xa, _ = rec.cross_attention(q, dt[:, :C], dt[:, C:], di[:, :C], di[:, C:], batch=B, heads=H, nq=n, nt=NT, nip=p, d=d, w_text=1.0, w_ip=1.0)
    unf = rec.gemm(xa, wo.cuda(), bias=bo.cuda(), residual=dhs, rows_per_image=n)
    rec.run(); torch.cuda.synchronize()
    o, u = out.float().cpu(), unf.float().cpu()
    bad = ~torch.isfinite(o) | ((o - u).abs() > 0.05 * (1 + u.abs()))
    print(f"n={n} p={p} ln={ln}: non-finite {(~torch.isfinite(o)).sum().item()}, bad {bad.sum().item()} of {o.numel()}")
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("  bad rows (first 40):", rows[:40].tolist(), "count", len(rows))
        print("  bad cols:", cols.tolist()[:80], "count", len(cols))
        r = rows[0].item()
        print("  row", r, "fused:", o[r, :16].tolist()); print("  row", r, "unfused:", u[r, :16].tolist())
