#!/usr/bin/env python3
"""K = 320 Linear (M = 65536, N = 320) of the 64 x 64 transformer blocks: the tiled GEMM (pv_gemm_conv) against the row-owning launch (pv_row_gemm), sustained."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.ops import Recorder  # noqa: E402
dev = torch.device("cuda")
M, K = 65536, 320
x = torch.randn(M, K, device=dev).half()
res = torch.randn(M, K, device=dev).half()


def sustained(rec, warm=500, reps=200):
    for _ in range(warm):
        rec.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rec.run()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for N in (320, 960):
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.zeros(N, device=dev)
    a, r, c = Recorder(dev), Recorder(dev), Recorder(dev)
    a.gemm(x, w, bias=b)
    r.row_gemm(x, w, bias=b)
    if N == 320:
        c.gemm(x, w, bias=b, residual=res)
    ta, tr = sustained(a), sustained(r)
    tc = sustained(c) if N == 320 else float("nan")
    byts = 2.0 * (M * K + M * N)
    print(f"N={N}: tiled {ta:6.1f} us ({byts / ta / 1e6:.2f} TB/s)   row-owning {tr:6.1f} us ({byts / tr / 1e6:.2f} TB/s)   tiled + residual {tc:6.1f} us", flush=True)
