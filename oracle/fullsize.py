"""TEST INFRASTRUCTURE: weights and inputs of the FULL-SIZE parity cases (SD-v1.5 UNet 859.5 M parameters, CLIP ViT-L/14, 12-layer text
encoder, 1024-wide adapters, SD VAE), shared by ``oracle/make_fullsize_golden.py`` - which runs the fp32 oracle on them in the BUILD
container and commits the expected tensors under ``tests/golden/full_*.pt`` - and by the ``-m gpu`` tests, which only CONSTRUCT the same
weights / inputs (seeded default inits: data, no oracle arithmetic) and compare the HIP path with the committed expectations.  The GPU box
therefore no longer spends minutes of host time inside the fp32 oracle; one live-oracle canary test remains.

Every builder seeds the global generator itself, so the order in which tests call them does not matter.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn as nn

EMPTY_PROMPT_IDS = [49406] + [49407] * 76          # CLIP tokenizer output for "" padded to 77 (what infer.py:86-88 feeds the uncond branch)


@contextlib.contextmanager
def no_init():
    """Skip the default parameter init of Linear / Conv / Embedding / norm layers while a model is constructed only to be loaded."""
    saved = []
    for cls in (nn.Linear, nn.Conv2d, nn.Embedding, nn.LayerNorm, nn.GroupNorm):
        saved.append((cls, cls.reset_parameters))
        cls.reset_parameters = lambda self: None
    try:
        yield
    finally:
        for cls, fn in saved:
            cls.reset_parameters = fn


def unet_state(seed: int = 0):
    """State dict of the oracle SD-v1.5 UNet with PhotoVerse processors under ``torch.manual_seed(seed)`` (torch default inits;
    ``num_tokens`` does not enter any shape)."""
    from oracle.unet_ref import UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
    torch.manual_seed(seed)
    ref = UNet2DConditionModelRef()
    set_visual_cross_attention_adapter_ref(ref, (5,))
    return {k: v.detach() for k, v in ref.state_dict().items()}


def text_state(seed: int = 2):
    from oracle.clip_ref import CLIPTextModelRef
    torch.manual_seed(seed)
    return {k: v.detach() for k, v in CLIPTextModelRef().state_dict().items()}


def vision_state(seed: int = 5):
    from oracle.clip_ref import CLIPVisionModelRef
    torch.manual_seed(seed)
    return {k: v.detach() for k, v in CLIPVisionModelRef().state_dict().items()}


def adapter_state(seed: int, tokens: int = 5):
    from oracle.adapters_ref import PhotoVerseAdapterRef
    torch.manual_seed(seed)
    return {k: v.detach() for k, v in PhotoVerseAdapterRef(1024, 768, tokens).state_dict().items()}


def vae_state(seed: int = 6):
    from oracle.vae_ref import AutoencoderKLDecoderRef
    torch.manual_seed(seed)
    return {k: v.detach() for k, v in AutoencoderKLDecoderRef(with_encoder=True).state_dict().items()}


# ------------------------------------------------------------------ inputs
def forward_case():
    g = torch.Generator().manual_seed(3)
    return dict(x=torch.randn(1, 4, 64, 64, generator=g), text=torch.randn(1, 77, 768, generator=g), ip=torch.randn(1, 1, 768, generator=g), t=481)


def cfg4_case():
    """BASELINE configs[4] per-rank shape: B = 4, 96 x 96 latents, P = 6 image tokens."""
    g = torch.Generator().manual_seed(44)
    return dict(x=torch.randn(4, 4, 96, 96, generator=g), text=torch.randn(4, 77, 768, generator=g), ip=torch.randn(4, 6, 768, generator=g), t=321,
                samples=(0, 3))


def loop_case():
    """The headline schedule: 50-step DPM-Solver++, guidance 7.5, B = 1, 64 x 64 latents, P = 1."""
    from oracle.infer_ref import draw_noise_ref
    g = torch.Generator().manual_seed(31)
    cond = (torch.randn(1, 77, 768, generator=g), torch.randn(1, 1, 768, generator=g))
    uncond = (torch.randn(1, 77, 768, generator=g), torch.randn(1, 1, 768, generator=g))
    return dict(cond=cond, uncond=uncond, noise=draw_noise_ref(1, 4, 64, seed=6), guidance=7.5, steps=50, checkpoints=(10, 50))


def pipeline_case():
    """Whole generation (infer.py:72-123): CLIP ViT-L/14 -> adapters (token_index 0) -> injected text encoder -> CFG loop -> VAE decode."""
    g = torch.Generator().manual_seed(4)
    return dict(example={"pixel_values": torch.zeros(1, 3, 512, 512), "pixel_values_clip": torch.randn(1, 3, 224, 224, generator=g),
                         "text_input_ids": torch.randint(0, 49000, (1, 77), generator=g), "concept_placeholder_idx": torch.tensor([[5]])},
                layers=[4, 8, 12, 16], guidance=7.5, steps=50, checkpoints=(8, 50), token_index=0, noise_seed=9,
                uncond_ids=torch.tensor([EMPTY_PROMPT_IDS]))


def vae_case():
    return dict(z=torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(2)),
                x=torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(7)) * 2 - 1)


TRAIN_LORA = dict(r=8, lora_alpha=1, target_modules=["attn2.to_q", "attn2.to_k", "attn2.to_v"])


def fill_lora_(model: nn.Module, seed: int = 1) -> nn.Module:
    """LoRA factors of the training case, drawn per parameter NAME (works on the oracle's and on the product's wrapped model alike):
    A ~ N(0, 1/r) like peft's gaussian init, B ~ N(0, 0.05) (B = 0 at init would zero dA)."""
    import zlib
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "lora_" not in name:
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
            std = 1.0 / TRAIN_LORA["r"] if "lora_A" in name else 0.05
            p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))
    return model


def train_case():
    g = torch.Generator().manual_seed(41)
    B, E, T, D = 1, 5, 257, 1024
    # forced grad-mode fusion draws per Transformer2DModel name: u = 0.5 -> (1, 1), every branch of every layer carries gradient,
    # except one text-only (u = 0.1) and one image-only (u = 0.9) layer
    forced = {"default": 0.5, "down_blocks.1.attentions.1": 0.1, "up_blocks.2.attentions.0": 0.9}
    return dict(noisy=torch.randn(B, 4, 64, 64, generator=g), noise=torch.randn(B, 4, 64, 64, generator=g), timesteps=torch.tensor([417]),
                ids=torch.randint(0, 49000, (B, 77), generator=g), pidx=torch.tensor([[4]]),
                embs=[torch.randn(B, T, D, generator=g).half() for _ in range(E)], forced=forced, E=E)


def subsample(t: torch.Tensor, keep: int = 4096) -> torch.Tensor:
    """At most ~``keep`` evenly strided elements of a tensor (all of a small one): what the training fixture stores per gradient."""
    flat = t.detach().flatten()
    return flat[::max(1, flat.numel() // keep)].clone()
