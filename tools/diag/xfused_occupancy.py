#!/usr/bin/env python3
"""Fused attn2 kernel (C = 320): is a workgroup latency-bound or do the two co-resident workgroups of a CU saturate shared resources?  The launch at 256 / 512 /
1024 workgroups (B = 8 / 16 / 32 samples of 4096 rows) with two workgroups per CU (default) and with ONE (PV_XF_LDS_PAD=4096: the second does not fit), sustained.
If one-per-CU takes about as long per workgroup ROUND as two-per-CU, the partners do not slow each other (latency-bound: more resident waves would help);
if it takes about half, the partners share saturated resources (throughput-bound: only de-phasing / less work helps).
usage (GPU box): python tools/diag/xfused_occupancy.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys, torch
sys.path.insert(0, %r)
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
torch.manual_seed(0)
n, H, d, NT, P = 4096, 8, 40, 77, 1
C = H * d
row = []
for B in (8, 16, 32):
    hs, kvt, kvip = torch.randn(B * n, C, device=dev).half(), torch.randn(B * NT, 2 * C, device=dev).half(), torch.randn(B * P, 2 * C, device=dev).half()
    wq, wo = (torch.randn(C, C, device=dev) * C ** -0.5).half(), (torch.randn(C, C, device=dev) * C ** -0.5).half()
    rec = Recorder(dev)
    vn = torch.zeros(B, H, P, device=dev)
    kimg, vimg = rec.xattn_pack_kv(kvt[:, :C], kvt[:, C:], kvip[:, :C], kvip[:, C:], batch=B, heads=H, d=d, nt=NT, nip=P, vnorm=vn)
    rec.run(); torch.cuda.synchronize()
    r2 = Recorder(dev)
    r2.cross_attention_fused(hs, wq, r2.pack_wo_for_fused(wo), torch.zeros(C, device=dev), kimg, vimg, batch=B, nq=n, heads=H, d=d, nt=NT, nip=P,
                             ln_gamma=torch.ones(C, device=dev), ln_beta=torch.zeros(C, device=dev), w_text=1.0, w_ip=1.0)
    for _ in range(500): r2.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300): r2.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 300 * 1e3
    row.append("%%4d workgroups %%6.1f us" %% (B * n // 128, us))
print("   ".join(row))
""" % ROOT
for pad in ("0", "4096"):
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, PV_XF_LDS_PAD=pad), capture_output=True, text=True)
    print("PV_XF_LDS_PAD=%-5s (%s per CU):  %s" % (pad, "two workgroups" if pad == "0" else "one workgroup", out.stdout.strip() or out.stderr[-300:]), flush=True)
