#!/usr/bin/env python3
"""conv 320->320 @64 (B=16): burst vs sustained timing, with and without the resnet epilogue (temb row + residual)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photoverse_amd.ops import Recorder
dev = torch.device("cuda"); B = 16; hw = 64; cin = cout = 320
w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).half()
bias = torch.zeros(cout, device=dev)
x = torch.randn(B * hw * hw, cin, device=dev).half()
res = torch.randn(B * hw * hw, cout, device=dev).half()
temb = torch.randn(B, cout, device=dev)
def timeit(rec, reps):
    rec.run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): rec.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for label, kw in (("plain", {}), ("+temb+residual", dict(rowadd=temb, rowadd_ld=cout, residual=res))):
    rec = Recorder(dev)
    rec.gemm(x, w, bias=bias, conv=dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw), **kw)
    for reps in (5, 50, 500, 3000):
        print(f"{label:16s} reps={reps:5d}: {timeit(rec, reps):7.1f} us")
