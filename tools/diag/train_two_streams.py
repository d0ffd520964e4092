#!/usr/bin/env python3
"""Prototype: one training iteration at bs = 16 as TWO concurrent half-batches (two TrainStep plans of 8 samples on two HIP streams, gradients summed -
data parallelism inside one GPU) against the one-plan iteration.  The inference loop gets ~20 % from running its two CFG forwards side by side."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from photoverse_amd.adapters import PhotoVerseAdapter  # noqa: E402
from photoverse_amd.clip import CLIPTextModel  # noqa: E402
from photoverse_amd.lora import LoraConfig, inject_adapter_in_model  # noqa: E402
from photoverse_amd.optim import AdamW  # noqa: E402
from photoverse_amd.train import TrainStep  # noqa: E402
from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
unet = UNet2DConditionModel()
set_visual_cross_attention_adapter(unet, (5,))
inject_adapter_in_model(LoraConfig(r=8, lora_alpha=1, lora_dropout=0.1), unet)
unet.to(dev)
for m in unet.modules():
    if hasattr(m, "lora_B"):
        m.lora_B["default"].weight.data.normal_(0, 0.02)
text_encoder = CLIPTextModel().to(dev)
text_adapter = PhotoVerseAdapter(1024, 768, 5).to(dev)
image_adapter = PhotoVerseAdapter(1024, 768, 5).to(dev)
B, S, REPS = 16, 64, 5
g = torch.Generator().manual_seed(99)
full = dict(noisy_latents=torch.randn(B, 4, S, S, generator=g).to(dev), noise=torch.randn(B, 4, S, S, generator=g).to(dev),
            timesteps=torch.randint(0, 1000, (B,), generator=g), text_input_ids=torch.randint(0, 49000, (B, 77), generator=g).to(dev),
            placeholder_idx=torch.full((B, 1), 5).to(dev), image_embeddings=[torch.randn(B, 257, 1024, generator=g).half().to(dev) for _ in range(5)])


def half(i):
    sl = slice(i * B // 2, (i + 1) * B // 2)
    return {k: ([t[sl] for t in v] if isinstance(v, list) else v[sl]) for k, v in full.items()}


def bench(fn):
    fn(); fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3


ts = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=S, w=S, n_tokens=5, grad_scale=4096.0, fusion_seed=1)
groups = ts.trainable_parameters()
params = [p for g_ in groups.values() for p in g_]
opt = AdamW(params, lr=1e-5, weight_decay=1e-2)


def one():
    ts.step(**full)
    opt.step(clip_groups=list(groups.values()), max_norm=1.0, grad_scale=ts.grad_scale)


print(f"one plan, bs 16:                       {bench(one):7.2f} ms per iteration")
del ts
torch.cuda.empty_cache()
for big in ("128", None):
    if big:
        os.environ["PV_CONV_BIG"] = big
    else:
        os.environ.pop("PV_CONV_BIG", None)
    halves = [TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B // 2, h=S, w=S, n_tokens=5, grad_scale=4096.0, fusion_seed=1 + i) for i in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    ins = [half(0), half(1)]

    def two():
        main = torch.cuda.current_stream()
        grads = []
        for t_, s_, x_ in zip(halves, streams, ins):
            s_.wait_stream(main)
            with torch.cuda.stream(s_):
                t_.step(**x_)
                grads.append([p.grad for p in params])
        for s_ in streams:
            main.wait_stream(s_)
        torch._foreach_add_(grads[0], grads[1])
        for p, g_ in zip(params, grads[0]):
            p.grad = g_
        opt.step(clip_groups=list(groups.values()), max_norm=1.0, grad_scale=2.0 * halves[0].grad_scale)     # mean over the two halves

    print(f"two plans of 8 on two streams (threshold {big or 256}): {bench(two):7.2f} ms per iteration")
    del halves
    torch.cuda.empty_cache()
