"""``run_inference`` with the reference's signature (``/root/reference/models/infer.py:7-123``).

Conditioning runs once per call on HIP kernels (CLIP ViT x2, adapters x3, text encoder x2, ``infer.py:76-96``); the
denoising loop (``:98-119``) is the graph-captured ``DenoiseLoop``; the VAE decode + clamp of ``:121-123`` runs on
``photoverse_amd.vae.AutoencoderKL`` (or any object with ``.decode`` / ``.config.scaling_factor``).  With ``vae=None`` the
function returns the final LATENTS (the value of ``latents`` after ``:119``).  ``from_noised_image`` (``:62-65``) starts from
``add_noise(vae.encode(pixel_values).latent_dist.sample() * scaling_factor, noise, t_0)``; the posterior sample is drawn on the
device generator (``torch.manual_seed(seed)`` seeds it like the reference's, the streams of two platforms never match bit for bit).
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from .pipeline import DenoiseLoop
from .scheduler import DPMSolverMultistepScheduler


#: captured loops kept per UNet (each holds two engines of static activations: ~4 GB at bs=16) - least recently used evicted
MAX_CACHED_LOOPS = 2


def _loop_for(unet, batch, latent_size, n_ip, steps, guidance, scheduler, training_mode=False, fusion_seed=0) -> DenoiseLoop:
    cache = unet.__dict__.setdefault("_denoise_loops", OrderedDict())
    key = (batch, latent_size, n_ip, steps, float(guidance), bool(training_mode), int(fusion_seed))
    loop = cache.pop(key, None)
    if loop is None or loop.unet_version != unet.__dict__.get("_pack_version", 0):
        loop = DenoiseLoop(unet, batch, latent_size, n_ip, steps, guidance, scheduler=scheduler, training_mode=training_mode,
                           fusion_seed=fusion_seed)
        loop.unet_version = unet.__dict__.get("_pack_version", 0)
    cache[key] = loop                               # most recently used last
    while len(cache) > MAX_CACHED_LOOPS:
        cache.popitem(last=False)
    return loop


def run_inference(example, tokenizer, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae, scheduler,
                  device, image_encoder_layers_idx, latent_size=64, guidance_scale=1, timesteps=100, token_index=0,
                  disable_tqdm=False, seed=None, from_noised_image=False, training_mode=False, *, noise=None):
    """Same 11 positional + 8 keyword arguments as the reference.  ``noise`` (keyword-only, new): a caller-drawn start noise
    ``(B, C, latent, latent)`` replacing the draw of ``infer.py:52-59`` - used by the batch-sharded pipeline, which draws the
    global batch once and hands each rank its slice."""
    if training_mode and torch.is_grad_enabled():
        # the reference back-propagates through the last denoising step (infer.py:99).  Here that differentiated call is a static
        # forward + backward plan, not a dynamic autograd graph: train.TrainStep(face_loss=..., vae=...) replays exactly this function
        # (conditioning with gradient, T - 1 steps without, the last step + decode inside the plan).  Under torch.no_grad() the FORWARD
        # semantics of the mode are available from this entry point: the last step's forwards draw the grad-mode branch fusion of every
        # cross-attention layer (attention_processor.py:413-420), on the device, inside the captured step.
        raise NotImplementedError("run_inference(training_mode=True) with autograd enabled: use photoverse_amd.train.TrainStep(face_loss=..., vae=...) - "
                                  "the differentiated form of this call - or call under torch.no_grad() for the forward semantics of the mode")
    device = torch.device(device)
    # infer.py:39-40 - the sampler is rebuilt from the loaded scheduler's config on every call
    sch = DPMSolverMultistepScheduler.from_config(scheduler.config)
    batch = example["pixel_values"].shape[0] if "pixel_values" in example else example["pixel_values_clip"].shape[0]

    uncond_input_ids = example.get("negative_text_input_ids", None)                      # :43-49
    if uncond_input_ids is None:
        uncond_input_ids = tokenizer([""] * batch, padding="max_length", max_length=tokenizer.model_max_length,
                                     return_tensors="pt").input_ids

    shape = (batch, unet.config.in_channels, latent_size, latent_size)                    # :52-59 noise on CPU, then moved
    if noise is not None:
        if tuple(noise.shape) != shape:
            raise ValueError(f"noise has shape {tuple(noise.shape)}, expected {shape}")
        noise = noise.to(device)
    elif seed is None:
        noise = torch.randn(shape).to(device)
    else:
        generator = torch.manual_seed(seed)
        noise = torch.randn(shape, generator=generator).to(device)

    if from_noised_image:                                                                 # :62-65
        if vae is None or not hasattr(vae, "encode"):
            raise NotImplementedError("from_noised_image needs a vae with .encode (photoverse_amd.vae.AutoencoderKL)")
        latents0 = vae.encode(example["pixel_values"].to(device)).latent_dist.sample().detach() * vae.config.scaling_factor
        sch.set_timesteps(timesteps)
        noise = sch.add_noise(latents0, noise, sch.timesteps[:1].repeat(latents0.shape[0]))     # :65

    placeholder_idx = example["concept_placeholder_idx"].to(device)                       # :72-73
    pixel_values_clip = example["pixel_values_clip"].to(device)

    image_features = image_encoder(pixel_values_clip, output_hidden_states=True)          # :76-78
    uncond_image_features = image_encoder(torch.zeros_like(pixel_values_clip), output_hidden_states=True)
    image_embeddings = [image_features[0]] + [image_features[2][i] for i in image_encoder_layers_idx if i < len(image_features[2])]
    uncond_image_embeddings = [uncond_image_features[0]] + [uncond_image_features[2][i] for i in image_encoder_layers_idx
                                                            if i < len(uncond_image_features[2])]

    concept_text_embeddings = text_adapter(image_embeddings, token_index=token_index)     # :89-91
    encoder_hidden_states_image = image_adapter(image_embeddings, token_index=token_index)
    uncond_encoder_hidden_states_image = image_adapter(uncond_image_embeddings, token_index=token_index)

    uncond_embeddings = text_encoder({"text_input_ids": uncond_input_ids.to(device)})[0]  # :93-96
    encoder_hidden_states = text_encoder({"text_input_ids": example["text_input_ids"].to(device),
                                          "concept_text_embeddings": concept_text_embeddings,
                                          "concept_placeholder_idx": placeholder_idx})[0]

    loop = _loop_for(unet, batch, latent_size, encoder_hidden_states_image.shape[1], timesteps, guidance_scale, sch,     # :98-119
                     training_mode=training_mode, fusion_seed=0 if seed is None else int(seed))
    loop.set_conditioning((encoder_hidden_states, encoder_hidden_states_image), (uncond_embeddings, uncond_encoder_hidden_states_image))
    loop.reset(noise)
    latents = loop.run().clone()

    if vae is None:
        return latents
    from .ops import Recorder
    rec = Recorder(latents.device)                                                        # :121: 1 / scaling_factor * latents
    inv = torch.full((latents.shape[0],), 1.0 / vae.config.scaling_factor, dtype=torch.float32, device=latents.device)
    _latents = rec.affine_rows(latents.contiguous(), inv)
    rec.run()
    images = vae.decode(_latents).sample                                                  # :122-123
    if images.is_cuda and images.dtype == torch.float32 and images.is_contiguous():
        rec = Recorder(images.device)
        rec.clamp_(images, -1.0, 1.0)
        rec.run()
        return images
    return images.clamp(-1, 1)
