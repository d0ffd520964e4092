"""Loaders and the checkpoint layout of ``/root/reference/models/modeling_utils.py:13-95``.

``load_models`` returns the same 9-tuple in the same order (``modeling_utils.py:95``; unpacked at ``generate.py:70``):
``tokenizer, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, lora_config``.

Differences forced by the environment (no network, no diffusers / peft): weights come either from a LOCAL directory in
the HF layout (``unet/diffusion_pytorch_model.safetensors``, ``text_encoder/model.safetensors``,
``vae/diffusion_pytorch_model.safetensors``, ``tokenizer/{vocab.json,merges.txt}`` - names are diffusers/transformers-compatible)
or from seeded random init (``pretrained_model_name_or_path=None`` or ``"random"``).  The reference pulls the image encoder
from a SEPARATE repository (``openai/clip-vit-large-patch14``, ``modeling_utils.py:59``): pass its local directory as
``image_encoder_path`` (default: ``<model dir>/image_encoder``).  Loading is LOUD: a missing weight file or parameters left
uninitialised after a load raise (``strict_load=True``, the default) instead of silently keeping random values.  The ``vae`` slot
holds ``photoverse_amd.vae.AutoencoderKL`` (encoder + decoder).
"""
from __future__ import annotations

import os
import warnings
from types import SimpleNamespace

import torch

from .adapters import PhotoVerseAdapter
from .clip import CLIPTextModel, CLIPVisionModel, patch_clip_text_transformer
from .lora import LoraConfig, inject_adapter_in_model
from .scheduler import DPMSolverMultistepScheduler
from .tokenizer import SyntheticCLIPTokenizer, load_tokenizer
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


#: deprecated diffusers attention parameter names still present in the published SD-v1.5 VAE weights
#: (``mid_block.attentions.0.{query,key,value,proj_attn}``); diffusers converts them on load, so do we
_DEPRECATED_ATTN = {".query.": ".to_q.", ".key.": ".to_k.", ".value.": ".to_v.", ".proj_attn.": ".to_out.0."}


def _rename_deprecated_vae_keys(sd):
    out = {}
    for k, v in sd.items():
        if ".attentions." in k:
            for old, new in _DEPRECATED_ATTN.items():
                if old in k:
                    k = k.replace(old, new)
                    if v.dim() == 4:                 # very old checkpoints store the projections as 1x1 convs
                        v = v.reshape(v.shape[0], v.shape[1])
                    break
        out[k] = v
    return out


def _strip_prefix(prefix):
    return lambda k: k[len(prefix):] if k.startswith(prefix) else k


def _load_safetensors_into(module, path, what, *, rename=None, ignore_missing=(), strict=True):
    """Load one safetensors file.  Returns (missing, unexpected); with ``strict`` any parameter of ``module`` the file does not
    provide raises (a half-loaded model would silently run on random weights)."""
    from safetensors.torch import load_file
    if not os.path.exists(path):
        msg = f"load_models: {what} weights not found at {path}"
        if strict:
            raise FileNotFoundError(msg + " (pass strict_load=False to keep seeded random-init weights for it)")
        warnings.warn(msg + "; keeping seeded random-init weights", stacklevel=3)
        return None, None
    sd = load_file(path)
    if rename is not None:
        sd = rename(sd)
    own = module.state_dict()
    missing, unexpected = module.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    unexpected = [k for k in sd if k not in own]
    missing = [k for k in missing if not any(pat in k for pat in ignore_missing)]
    if missing:
        msg = f"load_models: {len(missing)} parameter(s) of the {what} are not in {path} (first: {missing[:4]})"
        if strict:
            raise KeyError(msg)
        warnings.warn(msg + "; they keep their random-init values", stacklevel=3)
    if unexpected:
        warnings.warn(f"load_models: {len(unexpected)} tensor(s) in {path} have no counterpart in the {what} (first: {unexpected[:4]})",
                      stacklevel=3)
    return missing, unexpected


def _clip_vision_only(sd):
    """``openai/clip-vit-large-patch14`` ships the full CLIP model: keep the vision tower (``vision_model.*``), as
    ``CLIPVisionModel.from_pretrained`` does."""
    return {k: v for k, v in sd.items() if k.startswith("vision_model.")} or sd


def load_models(pretrained_model_name_or_path, extra_num_tokens, photoverse_path=None, use_lora=False, lora_config=None, *,
                seed=0, unet_config=None, vision_config=None, text_config=None, vae_config=None, image_encoder_path=None,
                strict_load=True):
    local = pretrained_model_name_or_path not in (None, "random") and os.path.isdir(str(pretrained_model_name_or_path))
    if pretrained_model_name_or_path not in (None, "random") and not local:
        raise FileNotFoundError(
            f"{pretrained_model_name_or_path!r} is not a local directory: this build has no network access; pass a local HF-layout "
            "directory or None / 'random' for seeded random-init weights")
    torch.manual_seed(seed)
    tokenizer = load_tokenizer(pretrained_model_name_or_path if local else None)
    if local and isinstance(tokenizer, SyntheticCLIPTokenizer):
        msg = (f"load_models: no tokenizer/vocab.json + merges.txt under {pretrained_model_name_or_path}: real weights would be "
               "driven by the synthetic (hash) tokenizer, whose ids mean nothing to them")
        if strict_load:
            raise FileNotFoundError(msg + " (pass strict_load=False to accept that)")
        warnings.warn(msg)
    text_encoder = CLIPTextModel(**(text_config or {}))
    vae = AutoencoderKL(**(vae_config or {}))
    unet = UNet2DConditionModel(**(unet_config or {}))
    image_encoder = CLIPVisionModel(**(vision_config or {}))
    scheduler = SimpleNamespace(config=DPMSolverMultistepScheduler().config)     # plays the DDPMScheduler of :60 (only .config is read)
    if local:
        root = str(pretrained_model_name_or_path)
        ie_root = str(image_encoder_path) if image_encoder_path is not None else os.path.join(root, "image_encoder")
        # the PhotoVerse processors (attn2.processor.to_k_ip / to_v_ip) are not part of a stock SD-v1.5 UNet file: they come from the
        # photoverse*.pt checkpoint (or stay at their init for training)
        _load_safetensors_into(unet, os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"), "UNet", strict=strict_load,
                               ignore_missing=(".processor.",))
        _load_safetensors_into(text_encoder, os.path.join(root, "text_encoder", "model.safetensors"), "text encoder", strict=strict_load,
                               ignore_missing=("position_ids",))
        _load_safetensors_into(vae, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"), "VAE", strict=strict_load,
                               rename=_rename_deprecated_vae_keys)
        _load_safetensors_into(image_encoder, os.path.join(ie_root, "model.safetensors"), "CLIP image encoder (openai/clip-vit-large-patch14)",
                               strict=strict_load, ignore_missing=("position_ids",), rename=_clip_vision_only)
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
