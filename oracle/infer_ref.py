"""Oracle: restatement of ``run_inference`` (``/root/reference/models/infer.py:7-123``).
TEST INFRASTRUCTURE.  Composition of the other oracle pieces, so its pin status
is theirs (UNet / scheduler: PARITY UNPINNED).

VAE encode/decode (``infer.py:62-68,121-123``) is out of scope for this build
(SURVEY.md section 8f row 1): the oracle returns the final *latents*, i.e. the
value of ``latents`` after the loop at ``infer.py:119``.
"""
import torch

from .scheduler_ref import DPMSolverMultistepRef


def draw_noise_ref(batch, channels, latent_size, seed=None):
    """``infer.py:52-59``: noise is drawn on CPU from the global generator
    (``torch.manual_seed`` returns it) and only then moved to the device."""
    if seed is None:
        return torch.randn((batch, channels, latent_size, latent_size))
    generator = torch.manual_seed(seed)
    return torch.randn((batch, channels, latent_size, latent_size), generator=generator)


def conditioning_ref(example, image_encoder, text_encoder, text_adapter, image_adapter, image_encoder_layers_idx,
                     token_index=0, uncond_input_ids=None):
    """``infer.py:72-96``: the once-per-call conditioning stack."""
    pv = example["pixel_values_clip"]
    feats = image_encoder(pv, output_hidden_states=True)
    ufeats = image_encoder(torch.zeros_like(pv), output_hidden_states=True)
    embs = [feats[0]] + [feats[2][i] for i in image_encoder_layers_idx if i < len(feats[2])]
    uembs = [ufeats[0]] + [ufeats[2][i] for i in image_encoder_layers_idx if i < len(ufeats[2])]
    concept = text_adapter(embs, token_index=token_index)
    ip = image_adapter(embs, token_index=token_index)
    uip = image_adapter(uembs, token_index=token_index)
    utext = text_encoder({"text_input_ids": uncond_input_ids})[0]
    text = text_encoder({"text_input_ids": example["text_input_ids"], "concept_text_embeddings": concept,
                         "concept_placeholder_idx": example["concept_placeholder_idx"]})[0]
    return (text, ip), (utext, uip)


@torch.no_grad()
def denoise_ref(unet, noise, cond, uncond, guidance_scale=1.0, timesteps=100, collect=None, max_steps=None):
    """``infer.py:39-40,70,98-119``: two UNet forwards per step (uncond, then
    cond, each at batch B), CFG combine, ``scheduler.step``.  ``max_steps`` (test
    aid): stop after that many steps of the ``timesteps``-step schedule."""
    sch = DPMSolverMultistepRef()
    sch.set_timesteps(timesteps)
    latents = noise * sch.init_noise_sigma
    for i, t in enumerate(sch.timesteps):
        if max_steps is not None and i >= max_steps:
            break
        x = sch.scale_model_input(latents, t)
        eps_u = unet(x, t, encoder_hidden_states=uncond).sample
        eps_c = unet(x, t, encoder_hidden_states=cond).sample
        eps = eps_u + guidance_scale * (eps_c - eps_u)
        latents = sch.step(eps, t, latents)
        if collect is not None:
            collect.append(latents.clone())
    return latents


def run_inference_ref(example, tokenizer, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae, scheduler,
                      device, image_encoder_layers_idx, latent_size=64, guidance_scale=1, timesteps=100, token_index=0,
                      disable_tqdm=False, seed=None, from_noised_image=False, training_mode=False):
    """Restatement of the WHOLE of ``infer.py:7-123`` (same 11 + 8 arguments), composed of the pieces above.  Pinned against the reference's
    own function executed over the same oracle models (``oracle/make_ref_golden.py: infer_golden`` -> ``tests/golden/ref_infer_golden.pt``,
    ``tests/test_reference_pins.py``).  ``scheduler`` is only read for ``.config`` (``infer.py:39``: the sampler is rebuilt per call)."""
    cfg = scheduler.config if isinstance(scheduler.config, dict) else dict(getattr(scheduler.config, "__dict__", {}))
    sch = DPMSolverMultistepRef(cfg.get("num_train_timesteps", 1000), cfg.get("beta_start", 0.00085), cfg.get("beta_end", 0.012),
                                cfg.get("steps_offset", 1))
    sch.set_timesteps(timesteps)                                                             # :39-40
    batch = example["pixel_values"].shape[0]
    uncond_input_ids = example.get("negative_text_input_ids", None)                          # :43-49
    if uncond_input_ids is None:
        uncond_input_ids = tokenizer([""] * batch, padding="max_length", max_length=tokenizer.model_max_length, return_tensors="pt").input_ids
    noise = draw_noise_ref(batch, unet.config.in_channels, latent_size, seed).to(device)     # :52-59
    if from_noised_image:                                                                    # :62-65
        latents = vae.encode(example["pixel_values"].to(device)).latent_dist.sample().detach()
        latents = latents * vae.config.scaling_factor
        latents = sch.add_noise(latents, noise, sch.timesteps[:1].repeat(latents.shape[0]))
    else:
        latents = noise
    latents = latents * sch.init_noise_sigma                                                 # :70
    ex = dict(example)
    ex["pixel_values_clip"] = example["pixel_values_clip"].to(device)
    ex["concept_placeholder_idx"] = example["concept_placeholder_idx"].to(device)
    cond, uncond = conditioning_ref(ex, image_encoder, text_encoder, text_adapter, image_adapter, image_encoder_layers_idx,   # :72-96
                                    token_index=token_index, uncond_input_ids=uncond_input_ids.to(device))
    n = len(sch.timesteps)
    for i, t in enumerate(sch.timesteps):                                                    # :98-119
        with torch.set_grad_enabled(training_mode and (i == n - 1)):
            x = sch.scale_model_input(latents, t)
            eps_u = unet(x, t, encoder_hidden_states=uncond).sample
            eps_c = unet(x, t, encoder_hidden_states=cond).sample
            eps = eps_u + guidance_scale * (eps_c - eps_u)
            latents = sch.step(eps, t, latents)
    _latents = 1 / vae.config.scaling_factor * latents.clone()                               # :121-123
    return vae.decode(_latents).sample.clamp(-1, 1)
