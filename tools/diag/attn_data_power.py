#!/usr/bin/env python3
"""Is the d = 40 self-attention launch power-capped?  The same launch (B = 16, 8 heads, N = 4096) on random, small random, zero and constant operands: the
instruction stream is identical, only the bits toggling in the MFMA / LDS / VGPR data paths differ."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.ops import Recorder
dev = torch.device("cuda")
B, n, d = 16, 4096, 40
C = 8 * d
for name, mk in (("randn", lambda: torch.randn(B * n, 3 * C, device=dev).half()), ("zeros", lambda: torch.zeros(B * n, 3 * C, device=dev).half()),
                 ("randn*0.1", lambda: (0.1 * torch.randn(B * n, 3 * C, device=dev)).half()), ("const 1", lambda: torch.ones(B * n, 3 * C, device=dev).half())):
    qkv = mk()
    rec = Recorder(dev)
    rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=B, heads=8, nq=n, nk=n, d=d)
    for _ in range(3):
        rec.run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        rec.run()
    b.record()
    torch.cuda.synchronize()
    print(f"{name:10s} {a.elapsed_time(b) / 20 * 1e3:7.1f} us")
