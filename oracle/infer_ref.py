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
