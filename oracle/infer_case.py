"""TEST INFRASTRUCTURE: the seeded tiny models / inputs / argument sets shared by ``oracle/make_ref_golden.py: infer_golden`` (which runs the
REFERENCE's ``run_inference`` over them in the build container) and by the tests (which run the oracle restatement on the CPU and the HIP
path on the GPU over the same numbers).  Only seeds and configs live here; nothing of /root/reference."""
import torch

from oracle.seeded import fill_state_
from oracle.unet_ref import TINY_CONFIG

VIS = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3, image_size=56, patch_size=14)
TXT = dict(vocab_size=1000, hidden_size=768, num_attention_heads=12, intermediate_size=512, num_hidden_layers=2, max_position_embeddings=77)
#: channel counts the HIP conv kernels tile (multiples of 128); one 2x level
VAE = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256), layers_per_block=1, norm_num_groups=32, scaling_factor=0.18215,
           with_encoder=True)
SEEDS = dict(unet=157, vision=158, text=159, image_adapter=160, text_adapter=161, vae=162)
LATENT, BATCH, NUM_TOKENS = 16, 2, 2
#: 9 >= the tiny ViT's 4 hidden states: dropped by the ``i < len(image_features[2])`` filter of infer.py:80-84
LAYERS_IDX = [1, 9]

#: keyword sets of the fixture cases.  ``global_seed``: torch.manual_seed(...) issued by the CALLER before the call (``seed=None`` draws the
#: noise from the global generator, infer.py:52-55); ``negative``: example carries ``negative_text_input_ids``.
CASES = {
    "default_guidance1":  dict(kw=dict(latent_size=LATENT, timesteps=3, seed=9), negative=False),
    "cfg7.5_full_neg":    dict(kw=dict(latent_size=LATENT, guidance_scale=7.5, timesteps=4, token_index="full", seed=11), negative=True),
    "global_generator":   dict(kw=dict(latent_size=LATENT, guidance_scale=3.0, timesteps=3, token_index=1, disable_tqdm=True), negative=False,
                               global_seed=123),
    "from_noised_image":  dict(kw=dict(latent_size=LATENT, guidance_scale=2.0, timesteps=3, seed=5, from_noised_image=True, disable_tqdm=True),
                               negative=False),
    "training_mode":      dict(kw=dict(latent_size=LATENT, guidance_scale=3.0, timesteps=3, token_index="full", seed=21, training_mode=True,
                                       disable_tqdm=True), negative=True),
}


class TokenizerStub:
    """What ``infer.py:44-49`` needs of a tokenizer: ``model_max_length`` and a call returning ``.input_ids``; records its arguments."""
    model_max_length = 77

    def __init__(self, bos=998, eos=999):
        self.bos, self.eos, self.calls = bos, eos, []

    def __call__(self, texts, padding=None, max_length=None, return_tensors=None):
        from types import SimpleNamespace
        self.calls.append((list(texts), padding, max_length, return_tensors))
        ids = torch.full((len(texts), max_length), self.eos, dtype=torch.int64)
        ids[:, 0] = self.bos
        return SimpleNamespace(input_ids=ids)


def example(negative: bool):
    g = torch.Generator().manual_seed(204)
    ex = {"pixel_values": torch.rand(BATCH, 3, 2 * LATENT, 2 * LATENT, generator=g) * 2 - 1,        # tiny VAE: one 2x level
          "pixel_values_clip": torch.randn(BATCH, 3, 56, 56, generator=g),
          "text_input_ids": torch.randint(0, 990, (BATCH, 77), generator=g),
          "concept_placeholder_idx": torch.tensor([[5], [3]])}
    if negative:
        ex["negative_text_input_ids"] = torch.randint(0, 990, (BATCH, 77), generator=g)
    return ex


def quiet_posterior_(vae):
    """Make ``latent_dist.sample()`` (infer.py:63) deterministic to 3e-7: the log-variance half of ``quant_conv`` is forced far below diffusers'
    clamp (-30), so std = exp(-15) and the platform-specific random stream of the draw cannot matter (the HIP path draws on the device)."""
    with torch.no_grad():
        lc = vae.config.latent_channels
        vae.quant_conv.weight[lc:].zero_()
        vae.quant_conv.bias[lc:].fill_(-100.0)
    return vae


def fill_all_(unet=None, image_encoder=None, text_encoder=None, image_adapter=None, text_adapter=None, vae=None):
    for key, m in (("unet", unet), ("vision", image_encoder), ("text", text_encoder), ("image_adapter", image_adapter),
                   ("text_adapter", text_adapter), ("vae", vae)):
        if m is not None:
            fill_state_(m, SEEDS[key])
    if vae is not None:
        quiet_posterior_(vae)


def oracle_models(processor_installer=None):
    """The oracle's tiny models with the fixture's weights.  ``processor_installer(unet, num_tokens)`` installs the attention processors
    (default: the oracle's restatement; the fixture generator passes the REFERENCE's own ``set_visual_cross_attention_adapter``)."""
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from oracle.clip_ref import CLIPTextModelRef, CLIPVisionModelRef
    from oracle.unet_ref import UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
    from oracle.vae_ref import AutoencoderKLDecoderRef
    torch.manual_seed(0)
    unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    (processor_installer or set_visual_cross_attention_adapter_ref)(unet, num_tokens=(NUM_TOKENS,))
    m = dict(unet=unet, image_encoder=CLIPVisionModelRef(**VIS).eval(), text_encoder=CLIPTextModelRef(**TXT).eval(),
             image_adapter=PhotoVerseAdapterRef(VIS["hidden_size"], 768, NUM_TOKENS).eval(),
             text_adapter=PhotoVerseAdapterRef(VIS["hidden_size"], 768, NUM_TOKENS).eval(), vae=AutoencoderKLDecoderRef(**VAE).eval())
    fill_all_(**m)
    return m
