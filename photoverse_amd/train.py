"""Forward half of one PhotoVerse training step (``/root/reference/train.py:466-516``) on the HIP kernels.

What a training iteration evaluates before ``accelerator.backward(loss)``: VAE-encode the pixels and sample the posterior
(``:473-474``), draw noise and per-sample timesteps and add the noise (``:477-484``), CLIP image features -> both adapters in FULL
mode (all ``extra_num_tokens + 1`` tokens, ``:487-502``), the dict-input text encoder with concept injection (``:497-499``), the UNet
in grad mode - per-sample timesteps ``(B,)``, every cross-attention layer drawing its branch fusion
(``attention_processor.py:413-420``; here on the device, ``pv_fusion_draw``) - and the three loss terms (``:509-516,541``):

    loss = mse(noise_pred, noise) + 0.01 * mean|concept_text_embeddings| + 0.001 * mean(to_v_ip_norm stack)   [+ 0.01 * face loss]

The BACKWARD (dX through every kernel, dW of the adapters / ``to_k_ip`` / ``to_v_ip`` / LoRA, AdamW) and the ArcFace face loss are not
built; this module is the forward they will hang off, and what ``bench.py``'s ``train_forward`` object times.
"""
from __future__ import annotations

from typing import Optional

import torch

from .ops import Recorder


def _train_engine(unet, batch, h, w, n_ip, fusion_seed):
    cache = unet.__dict__.setdefault("_train_engines", {})
    key = (batch, h, w, n_ip, int(fusion_seed), unet.__dict__.get("_pack_version", 0))
    eng = cache.get(key)
    if eng is None:
        cache.clear()                                    # one training shape at a time: an engine holds ~2 GB of activations at bs=16
        eng = cache[key] = unet.engine(batch, h, w, n_ip, batch, device_fusion="always", fusion_seed=fusion_seed)
    return eng


@torch.no_grad()
def training_step_forward(batch, tokenizer, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae, noise_scheduler, device,
                          image_encoder_layers_idx, extra_num_tokens: int, *, generator: Optional[torch.Generator] = None, fusion_seed: int = 0,
                          forced_fusion=None, noise=None, timesteps=None, posterior_eps=None):
    """Returns a dict: ``loss``, ``diffusion_loss``, ``concept_text_loss``, ``cross_attn_visual_loss`` (fp32 device scalars, shape
    (1,)), ``noise_pred`` and the draws it used (``noise``, ``timesteps``, ``latents``).  ``noise`` / ``timesteps`` / ``posterior_eps``
    / ``forced_fusion`` (per cross-attention layer u in [0, 1), in ``engine.fusion_names`` order) override the random draws (tests)."""
    device = torch.device(device)
    pixel_values = batch["pixel_values"].to(device, dtype=torch.float32)
    pixel_values_clip = batch["pixel_values_clip"].to(device, dtype=torch.float32)
    placeholder_idx = batch["concept_placeholder_idx"].to(device)
    text_input_ids = batch["text_input_ids"].to(device)
    bsz = pixel_values.shape[0]

    # train.py:473-474 - latents = vae.encode(x).latent_dist.sample() * scaling_factor
    dist = vae.encode(pixel_values).latent_dist
    rec = Recorder(device)
    if posterior_eps is None:
        posterior_eps = torch.randn(dist.mean.shape, generator=generator, device="cpu" if generator is not None else device)
    sample = rec.posterior_sample(dist.parameters.contiguous(), posterior_eps.to(device, torch.float32).contiguous())
    sf = torch.full((bsz,), float(vae.config.scaling_factor), dtype=torch.float32, device=device)
    latents = rec.affine_rows(sample, sf)
    rec.run()

    # train.py:477-484 - noise, per-sample timesteps, forward diffusion
    if noise is None:
        noise = torch.randn(latents.shape, generator=generator, device="cpu" if generator is not None else device)
    noise = noise.to(device, torch.float32).contiguous()
    n_train = noise_scheduler.config["num_train_timesteps"] if isinstance(noise_scheduler.config, dict) else noise_scheduler.config.num_train_timesteps
    if timesteps is None:
        timesteps = torch.randint(0, n_train, (bsz,), generator=generator)
    timesteps = timesteps.long().cpu()
    from .scheduler import DPMSolverMultistepScheduler
    sch = noise_scheduler if hasattr(noise_scheduler, "add_noise") else DPMSolverMultistepScheduler.from_config(noise_scheduler.config)
    noisy_latents = sch.add_noise(latents, noise, timesteps)

    # train.py:487-502 - image features -> adapters (all tokens), text encoder with the injected concept embeddings
    image_features = image_encoder(pixel_values_clip, output_hidden_states=True)
    image_embeddings = [image_features[0]] + [image_features[2][i] for i in image_encoder_layers_idx if i < len(image_features[2])]
    assert len(image_embeddings) == extra_num_tokens + 1, "Entered indices are out of range for image_encoder layers."
    concept_text_embeddings = text_adapter(image_embeddings)
    encoder_hidden_states = text_encoder({"text_input_ids": text_input_ids, "concept_text_embeddings": concept_text_embeddings,
                                          "concept_placeholder_idx": placeholder_idx})[0]
    encoder_hidden_states_image = image_adapter(image_embeddings)

    # train.py:505-506 - the UNet in grad mode: per-sample timesteps, per-layer branch fusion drawn on the device
    B, _, h, w = noisy_latents.shape
    P = encoder_hidden_states_image.shape[1]
    eng = _train_engine(unet, B, h, w, P, fusion_seed)
    eng.x_in.copy_(noisy_latents)
    eng.text.copy_(encoder_hidden_states.reshape(-1, encoder_hidden_states.shape[-1]))
    eng.ip.copy_(encoder_hidden_states_image.reshape(-1, encoder_hidden_states_image.shape[-1]))
    eng.timesteps.copy_(timesteps.to(device, torch.float32))
    if forced_fusion is not None:
        eng.fusion_forced.copy_(torch.as_tensor(forced_fusion, dtype=torch.float32))
    else:
        eng.fusion_forced.fill_(-1.0)
    noise_pred = eng.run()

    # train.py:509-516, 541 - regularisers and the loss
    rec = Recorder(device)
    concept_text_loss = rec.reduce_mean(concept_text_embeddings.contiguous(), mode="abs")
    per_layer = rec.empty((len(eng.vnorms),), torch.float32)
    for i, name in enumerate(eng.vnorms):                       # get_visual_cross_attention_values_norm(unet).mean(): equal-sized layers
        rec.reduce_mean(eng.vnorms[name], mode="mean", out=per_layer[i:i + 1])
    cross_attn_visual_loss = rec.reduce_mean(per_layer, mode="mean")
    diffusion_loss = rec.reduce_mean(noise_pred.contiguous(), noise, mode="mse")
    terms = rec.hold(torch.zeros(3, dtype=torch.float32, device=device))
    rec.run()
    # loss = diffusion + 0.01 * concept + 0.001 * visual: three scalars, combined by the same row-affine kernel
    terms[0:1].copy_(diffusion_loss); terms[1:2].copy_(concept_text_loss); terms[2:3].copy_(cross_attn_visual_loss)
    rec2 = Recorder(device)
    w = torch.tensor([3.0, 0.03, 0.003], dtype=torch.float32, device=device)     # mean of the three weighted terms == 1 * d + 0.01 * c + 0.001 * v
    weighted = rec2.affine_rows(terms.view(3, 1), w)
    loss = rec2.reduce_mean(weighted.view(-1), mode="mean")
    rec2.run()
    return {"loss": loss, "diffusion_loss": diffusion_loss, "concept_text_loss": concept_text_loss,
            "cross_attn_visual_loss": cross_attn_visual_loss, "noise_pred": noise_pred, "noise": noise, "timesteps": timesteps,
            "latents": latents, "fusion_table": eng.fusion_tab.clone(), "fusion_names": list(eng.fusion_names)}
