#!/usr/bin/env python3
"""Per-shape timing of ONE launch plan of a denoising step (GPU box): every (symbol, GFLOP, workgroups) group of tagged launches replayed alone between
HIP events.  usage: plan_breakdown.py [merged|head|tail] [B S P]   (default 16 64 1 = configs[1]; 4 96 6 = configs[4] per rank)"""
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_random_unet  # noqa: E402
from photoverse_amd.pipeline import DenoiseLoop  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "merged"
B, S, P = (int(a) for a in sys.argv[2:5]) if len(sys.argv) >= 5 else (16, 64, 1)
dev = torch.device("cuda")
unet = build_random_unet(P, dev)
loop = DenoiseLoop(unet, B, S, P, 50, 7.5)
g = torch.Generator().manual_seed(0)
loop.set_conditioning((torch.randn(B, 77, 768, generator=g).to(dev), torch.randn(B, P, 768, generator=g).to(dev)),
                      (torch.randn(B, 77, 768, generator=g).to(dev), torch.randn(B, P, 768, generator=g).to(dev)))
loop.reset(torch.randn(B, 4, S, S, generator=g))
loop.step()
torch.cuda.synchronize()
rec = {"merged": loop.engines_m[0].rec, "head": loop.engines_u[0].rec_head, "tail": loop.engines_u[0].rec_tail}[which]
groups = OrderedDict()
for t in rec.tags:
    groups.setdefault((t[0], round(t[1] / 1e9, 2), t[3] if len(t) > 3 else None), 0)
    groups[(t[0], round(t[1] / 1e9, 2), t[3] if len(t) > 3 else None)] += 1
rows = []
for (name, gf, wgs), n in groups.items():
    sub = rec.subset(lambda t, name=name, gf=gf, wgs=wgs: t[0] == name and round(t[1] / 1e9, 2) == gf and (t[3] if len(t) > 3 else None) == wgs)
    for _ in range(3):
        sub.run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        sub.run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    rows.append((ms, n, gf, wgs, name))
tot = sum(r[0] for r in rows)
print(f"{which} plan: {len(rec.tags)} tagged launches, {tot:.3f} ms summed over the groups (each group replayed alone)")
for ms, n, gf, wgs, name in sorted(rows, reverse=True):
    tf = gf * n / ms if ms > 0 else 0.0
    print(f"  {ms:7.3f} ms  {n:3d} x {ms / n * 1e3:7.1f} us  {gf:8.2f} GFLOP  {tf:7.1f} TFLOP/s  wgs={wgs}  {name}")
