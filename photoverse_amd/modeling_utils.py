"""Loaders and the checkpoint layout of ``/root/reference/models/modeling_utils.py:13-95``.

``load_models`` returns the same 9-tuple in the same order (``modeling_utils.py:95``; unpacked at ``generate.py:70``):
``tokenizer, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, lora_config``.

Differences forced by the environment (no network, no diffusers / peft): weights come either from a LOCAL directory in
the HF layout (``unet/diffusion_pytorch_model.safetensors``, ``text_encoder/model.safetensors``,
``image_encoder/model.safetensors``, ``vae/diffusion_pytorch_model.safetensors`` - names are diffusers/transformers-compatible)
or from seeded random init (``pretrained_model_name_or_path=None`` or ``"random"``); the ``vae`` slot holds the DECODER half
only (``photoverse_amd.vae.AutoencoderKL``: ``decode`` exists, ``encode`` does not, so ``from_noised_image`` needs a
caller-supplied VAE).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch

from .adapters import PhotoVerseAdapter
from .clip import CLIPTextModel, CLIPVisionModel, patch_clip_text_transformer
from .lora import LoraConfig, inject_adapter_in_model
from .scheduler import DPMSolverMultistepScheduler
from .tokenizer import SyntheticCLIPTokenizer
from .unet import UNet2DConditionModel, set_visual_cross_attention_adapter
from .vae import AutoencoderKL


def load_photoverse_model(path, image_adapter, text_adapter, unet):
    """``modeling_utils.py:13-26``: optional lora_config is applied BEFORE the weights are loaded; ``strict=False`` for the
    UNet subset; the ``optimizer`` entry is ignored (it is written but never read by the reference either)."""
    state_dict = torch.load(path, map_location="cpu")
    lora_config = None
    if "lora_config" in state_dict:
        lora_config = LoraConfig(**{k: v for k, v in state_dict["lora_config"].items() if k in LoraConfig.__dataclass_fields__})
        unet = inject_adapter_in_model(lora_config, unet)
    if "image_adapter" in state_dict:
        image_adapter.load_state_dict(state_dict["image_adapter"])
    if "text_adapter" in state_dict:
        text_adapter.load_state_dict(state_dict["text_adapter"])
    if "cross_attention_adapter" in state_dict:
        unet.load_state_dict(state_dict["cross_attention_adapter"], strict=False)
    return image_adapter, text_adapter, unet, lora_config


def _unwrap(accelerator, m):
    return accelerator.unwrap_model(m) if accelerator is not None else m


def save_progress(image_adapter, text_adapter, unet, accelerator, output_path, step=None, lora_config=None, optimizer=None):
    """``modeling_utils.py:29-50``: same keys, same key filter, same file names."""
    state_dict_cross_attention = {}
    for key, value in _unwrap(accelerator, unet).state_dict().items():
        if "attn2" in key:
            if "processor" in key or "to_q" in key or "to_k" in key or "to_v" in key:
                state_dict_cross_attention[key] = value
    final_state_dict = {
        "image_adapter": _unwrap(accelerator, image_adapter).state_dict(),
        "text_adapter": _unwrap(accelerator, text_adapter).state_dict(),
        "cross_attention_adapter": state_dict_cross_attention,
    }
    if optimizer is not None:
        final_state_dict["optimizer"] = optimizer.state_dict()
    if lora_config is not None:
        final_state_dict["lora_config"] = lora_config.to_dict()
    name = f"photoverse_{str(step).zfill(6)}.pt" if step is not None else "photoverse.pt"
    torch.save(final_state_dict, os.path.join(output_path, name))


def _load_safetensors_into(module, path, prefix_fix=None):
    from safetensors.torch import load_file
    sd = load_file(path)
    if prefix_fix is not None:
        sd = {prefix_fix(k): v for k, v in sd.items()}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    return missing, unexpected


def load_models(pretrained_model_name_or_path, extra_num_tokens, photoverse_path=None, use_lora=False, lora_config=None, *,
                seed=0, unet_config=None, vision_config=None, text_config=None, vae_config=None):
    local = pretrained_model_name_or_path not in (None, "random") and os.path.isdir(str(pretrained_model_name_or_path))
    if pretrained_model_name_or_path not in (None, "random") and not local:
        raise FileNotFoundError(
            f"{pretrained_model_name_or_path!r} is not a local directory: this build has no network access; pass a local HF-layout "
            "directory or None / 'random' for seeded random-init weights")
    torch.manual_seed(seed)
    tokenizer = SyntheticCLIPTokenizer()
    text_encoder = CLIPTextModel(**(text_config or {}))
    vae = AutoencoderKL(**(vae_config or {}))
    unet = UNet2DConditionModel(**(unet_config or {}))
    image_encoder = CLIPVisionModel(**(vision_config or {}))
    scheduler = SimpleNamespace(config=DPMSolverMultistepScheduler().config)     # plays the DDPMScheduler of :60 (only .config is read)
    if local:
        root = str(pretrained_model_name_or_path)
        for mod, rel in ((unet, "unet/diffusion_pytorch_model.safetensors"), (text_encoder, "text_encoder/model.safetensors"),
                         (image_encoder, "image_encoder/model.safetensors"), (vae, "vae/diffusion_pytorch_model.safetensors")):
            f = os.path.join(root, rel)
            if os.path.exists(f):
                _load_safetensors_into(mod, f)
    for m in (unet, vae, text_encoder, image_encoder):                           # :63-66
        m.requires_grad_(False)
    image_adapter = PhotoVerseAdapter(cross_attention_dim=unet.config.cross_attention_dim,
                                      clip_embedding_dim=image_encoder.config.hidden_size, num_tokens=extra_num_tokens + 1)
    text_adapter = PhotoVerseAdapter(cross_attention_dim=unet.config.cross_attention_dim,
                                     clip_embedding_dim=image_encoder.config.hidden_size, num_tokens=extra_num_tokens + 1)
    text_encoder = patch_clip_text_transformer(text_encoder)                     # :81
    unet = set_visual_cross_attention_adapter(unet, num_tokens=(extra_num_tokens + 1,))   # :84
    if use_lora:
        assert lora_config is not None, "Lora config is required when using lora"
        unet = inject_adapter_in_model(lora_config, unet)
    if photoverse_path is not None:
        image_adapter, text_adapter, unet, lora_config = load_photoverse_model(photoverse_path, image_adapter, text_adapter, unet)
    return tokenizer, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, lora_config
