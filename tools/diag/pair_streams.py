#!/usr/bin/env python3
"""Prototype: a step as TWO INDEPENDENT sample groups instead of two CFG branches.  Group g = half of the samples; its uncond and cond forwards run as ONE
batch-B plan (rows [uncond(group); cond(group)]: weights streamed once, same M per launch as today's per-branch plans) followed by its own CFG + solver
step.  The groups never meet: each replays its own graph on its own stream, so the two streams drift to different depths of the UNet instead of running the
same kernel at the same time, and no plan runs alone.  Prints steps/s next to the shipped DenoiseLoop on the same box.  usage: pair_streams.py [skew_ms]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import build_random_unet  # noqa: E402
from photoverse_amd.ops import Recorder  # noqa: E402
from photoverse_amd.pipeline import DenoiseLoop  # noqa: E402
from photoverse_amd.scheduler import DPMSolverMultistepScheduler  # noqa: E402

B, S, P, T, STEPS, WARM = 16, 64, 1, 50, 40, 6
dev = torch.device("cuda")
unet = build_random_unet(P, dev)
cfgm = unet.config
g = torch.Generator().manual_seed(0)
cond = (torch.randn(B, 77, 768, generator=g).to(dev), torch.randn(B, P, 768, generator=g).to(dev))
unc = (torch.randn(B, 77, 768, generator=g).to(dev), torch.randn(B, P, 768, generator=g).to(dev))
noise = torch.randn(B, 4, S, S, generator=g)


class Group:
    def __init__(self, sl, big_min):
        b = sl.stop - sl.start
        self.b = b
        sch = DPMSolverMultistepScheduler()
        sch.set_timesteps(T)
        self.sch = sch
        f32, f16 = torch.float32, torch.float16
        self.lat2 = torch.zeros((2 * b, 4, S, S), dtype=f32, device=dev)          # [uncond rows; cond rows]: the same latents twice
        self.latents = self.lat2[:b]
        self.x0_prev = torch.zeros((b, 4, S, S), dtype=f32, device=dev)
        self.timesteps = sch.timesteps.to(device=dev, dtype=f32)
        self.coef = sch.coefficient_table().to(dev)
        self.state = torch.tensor([0, T, 0, 0], dtype=torch.int32, device=dev)
        self.text = torch.cat([unc[0][sl], cond[0][sl]]).reshape(2 * b * 77, 768).half().contiguous()
        self.ip = torch.cat([unc[1][sl], cond[1][sl]]).reshape(2 * b * P, 768).half().contiguous()
        self.eps = torch.empty((2 * b, 4, S, S), dtype=f32, device=dev)
        kw = dict(big_min=big_min) if big_min else {}
        self.eng = unet.engine(2 * b, S, S, P, 1, latents_in=self.lat2, text=self.text, ip=self.ip, out=self.eps, timesteps=self.timesteps, state=self.state,
                               n_text=77, **kw)
        self.tail = Recorder(dev)
        self.tail.cfg_dpm_step(self.eps[:b], self.eps[b:], self.latents, self.x0_prev, self.coef, self.state, 7.5)
        self.tail.step_advance(self.state)
        self.eng.run_conditioning()
        self.latents.copy_(noise[sl].to(dev) * sch.init_noise_sigma)
        self.stream = torch.cuda.Stream(device=dev)
        self.graph = None

    def eager(self):
        self.lat2[self.b:].copy_(self.lat2[:self.b])
        self.eng.rec.run()
        self.tail.run()

    def capture(self):
        torch.cuda.synchronize()
        with torch.cuda.stream(self.stream):
            keep = (self.lat2.clone(), self.x0_prev.clone(), self.state.clone())
            self.eager()
            self.lat2.copy_(keep[0]); self.x0_prev.copy_(keep[1]); self.state.copy_(keep[2])
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=self.stream):
            self.eager()
        self.graph = gr

    def step(self):
        with torch.cuda.stream(self.stream):
            self.graph.replay()


def run_groups(groups, steps, skew_ms):
    torch.cuda.synchronize()
    if skew_ms > 0:                      # the second group starts `skew_ms` late (a spin kernel on its stream)
        with torch.cuda.stream(groups[1].stream):
            torch.cuda._sleep(int(skew_ms * 1e-3 * 2.0e9))
    t0 = time.perf_counter()
    for _ in range(steps):
        for gr in groups:
            gr.step()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


skews = [float(a) for a in sys.argv[1:]] or [0.0, 8.0]
for big_min in (128, 256):
    groups = [Group(slice(0, B // 2), big_min), Group(slice(B // 2, B), big_min)]
    for gr in groups:
        gr.capture()
    for skew in skews:
        run_groups(groups, WARM, 0.0)
        dt = run_groups(groups, STEPS, skew)
        fin = all(torch.isfinite(gr.latents).all().item() for gr in groups)
        print(f"two sample groups, big_min {big_min}, initial skew {skew:4.1f} ms: {STEPS / dt:6.2f} steps/s ({dt / STEPS * 1e3:.3f} ms / step incl. the skew), "
              f"launches per step {2 * (len(groups[0].eng.rec) + len(groups[0].tail) + 1)}, finite {fin}")
        for gr in groups:                # back to step 0 (T rows in the tables)
            gr.state.copy_(torch.tensor([0, T, 0, 0], dtype=torch.int32, device=dev))
    del groups
    torch.cuda.empty_cache()

loop = DenoiseLoop(unet, B, S, P, T, 7.5)
loop.set_conditioning(cond, unc)
loop.reset(noise)
for _ in range(WARM):
    loop.step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    loop.step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"shipped DenoiseLoop (three plans, two branches): {STEPS / dt:6.2f} steps/s ({dt / STEPS * 1e3:.3f} ms / step), launches per step {loop.launches_per_step}")
