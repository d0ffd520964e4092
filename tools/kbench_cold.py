#!/usr/bin/env python3
"""conv 320->320 @64 (B=16) with hot vs cold inputs: cycle through NBUF distinct input/output buffers (> Infinity Cache)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from photoverse_amd.ops import Recorder
dev = torch.device("cuda"); B = 16; hw = 64; cin = cout = 320
w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).half()
bias = torch.zeros(cout, device=dev)
for nbuf in (1, 4, 16):
    recs = []
    for i in range(nbuf):
        rec = Recorder(dev)
        x = torch.randn(B * hw * hw, cin, device=dev).half()
        rec.gemm(x, w, bias=bias, conv=dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw))
        recs.append(rec)
    for r in recs: r.run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        for r in recs: r.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * nbuf)
    print(f"nbuf={nbuf:3d} ({nbuf * 84} MB footprint): {us:7.1f} us per conv")
    # producer-consumer: a streaming write of the input right before the conv (like gn_apply does)
    rec = recs[0]
    src = torch.randn(B * hw * hw, cin, device=dev).half()
    dst = rec.keep[1] if False else None
xs = [torch.randn(B * hw * hw, cin, device=dev).half() for _ in range(2)]
rec = Recorder(dev); xin = torch.empty_like(xs[0]); rec.gemm(xin, w, bias=bias, conv=dict(batch=B, hin=hw, win=hw, hout=hw, wout=hw))
big = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)
for label, thrash in (("copy->conv", False), ("copy->300MB memset->conv", True)):
    tot = 0.0
    for i in range(6):
        xin.copy_(xs[i % 2])
        if thrash: big.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rec.run(); e1.record(); torch.cuda.synchronize()
        if i >= 2: tot += e0.elapsed_time(e1) * 1e3
    print(f"{label}: {tot / 4:7.1f} us per conv")
