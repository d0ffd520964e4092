#!/usr/bin/env python3
"""What one denoising step launches (GPU box): the recorded tags of the default DenoiseLoop aggregated by (kernel, flops, workgroups) per plan."""
import os
import sys
from collections import Counter

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_random_unet  # noqa: E402
from photoverse_amd.pipeline import DenoiseLoop  # noqa: E402

B, S, P = 16, 64, 1
dev = torch.device("cuda")
unet = build_random_unet(P, dev)
loop = DenoiseLoop(unet, B, S, P, 50, 7.5)
names = {id(e): n for n, lst in (("uncond", loop.engines_u), ("cond", loop.engines_c), ("merged", loop.engines_m)) for e in lst}
for e in loop.all_engines:
    c = Counter((t[0], round(t[1] / 1e9, 2), t[3] if len(t) > 3 else None) for t in e.rec.tags)
    print(f"== {names[id(e)]} plan, batch {e.B}: {len(e.rec.tags)} tagged launches")
    for (k, gf, wgs), n in sorted(c.items(), key=lambda kv: -kv[0][1] * kv[1]):
        print(f"  {n:3d} x {gf:9.2f} GFLOP  wgs={wgs}  {k}")
