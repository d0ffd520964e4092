#!/usr/bin/env python3
"""How a denoising step's time splits over its launch plans (GPU box): the uncond / cond heads side by side, the merged low-resolution plan, the tails side
by side - each replayed alone between HIP events, with its algorithmic flops from the launch tags."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_random_unet  # noqa: E402
from photoverse_amd.pipeline import DenoiseLoop  # noqa: E402

B, S, P = 16, 64, 1
dev = torch.device("cuda")
unet = build_random_unet(P, dev)
loop = DenoiseLoop(unet, B, S, P, 50, 7.5)
g = torch.Generator().manual_seed(0)
loop.set_conditioning((torch.randn(B, 77, 768, generator=g).to(dev), torch.randn(B, P, 768, generator=g).to(dev)),
                      (torch.randn(B, 77, 768, generator=g).to(dev), torch.randn(B, P, 768, generator=g).to(dev)))
loop.reset(torch.randn(B, 4, S, S, generator=g))
main, side = torch.cuda.current_stream(), torch.cuda.Stream()


def timed(lanes, reps=10):
    def once():
        for sd, lane in zip((side,), lanes[1:]):
            sd.wait_stream(main)
            with torch.cuda.stream(sd):
                for r in lane:
                    r.run()
        for r in lanes[0]:
            r.run()
        if len(lanes) > 1:
            main.wait_stream(side)
    once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(reps):
        once()
    b.record(main)
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def fl(recs):
    return sum(t[1] for r in recs for t in r.tags) / 1e12


eu, ec, em = loop.engines_u[0], loop.engines_c[0], loop.engines_m[0]
rows = [("heads (uncond || cond)", [[eu.rec_head], [ec.rec_head]]), ("merged 16x16 / 8x8 / mid (batch 32)", [[em.rec]]), ("tails (uncond || cond)", [[eu.rec_tail], [ec.rec_tail]]),
        ("heads, one after the other", [[eu.rec_head, ec.rec_head]]), ("tails, one after the other", [[eu.rec_tail, ec.rec_tail]])]
tot = 0.0
for name, lanes in rows:
    ms = timed(lanes)
    f = fl([r for lane in lanes for r in lane])
    n = sum(len(r.tags) for lane in lanes for r in lane)
    print(f"{name:40s} {ms:7.3f} ms  {f:6.2f} TFLOP  {f / ms:6.3f} PFLOP/s = {f / ms / 2.5:5.3f} of peak  ({n} tagged launches)")
