"""Runs the fp32 oracle on the FULL-SIZE cases of ``oracle/fullsize.py`` and writes ``tests/golden/full_*.pt`` (expected tensors only;
weights and inputs are rebuilt from seeds by the tests).  BUILD container, ~25 min on 8 cores: ``python -m oracle.make_fullsize_golden
[case ...]`` with cases ``unet loop pipeline vae train tiny50`` (default: all).

  full_unet.pt      one forward at B = 1 / 64 x 64 (also the live-oracle canary of the GPU suite) and the configs[4] per-rank shape
                    (B = 4, 96 x 96, P = 6; samples 0 and 3)
  full_loop.pt      latents after 10 and after all 50 steps of the headline schedule (guidance 7.5)
  full_pipeline.pt  CLIP ViT-L/14 hidden states (sub-sampled), conditioning tensors, latents after 8 and 50 steps, decoded image (fp16)
  full_vae.pt       decoder 64 x 64 -> 512 x 512 (fp16, every 2nd pixel) and encoder 256 x 256 -> posterior mean / logvar
  full_train.pt     training step (train.py:466-516) at B = 1: loss, per-tensor gradient norms and strided sub-samples of every trainable
                    gradient (to_k_ip / to_v_ip, LoRA A / B with the INDEPENDENT peft restatement oracle/lora_ref.py, both adapters)
"""
import os
import sys
import time

import torch
import torch.nn.functional as F

from oracle import fullsize as fs

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
_cache = {}


def _unet():
    if "unet" not in _cache:
        from oracle.unet_ref import UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
        with fs.no_init():
            ref = UNet2DConditionModelRef().eval()
            set_visual_cross_attention_adapter_ref(ref, (5,))
        ref.load_state_dict(fs.unet_state())
        _cache["unet"] = ref
    return _cache["unet"]


def unet():
    ref = _unet()
    c = fs.forward_case()
    out = {}
    with torch.no_grad():
        out["forward_eps"] = ref(c["x"], torch.tensor(c["t"]), encoder_hidden_states=(c["text"], c["ip"])).sample
        c4 = fs.cfg4_case()
        out["cfg4_eps"] = {i: ref(c4["x"][i:i + 1], torch.tensor(c4["t"]), encoder_hidden_states=(c4["text"][i:i + 1], c4["ip"][i:i + 1])).sample
                           for i in c4["samples"]}
    torch.save(out, os.path.join(OUT, "full_unet.pt"))


def loop():
    from oracle.infer_ref import denoise_ref
    c = fs.loop_case()
    lat = []
    denoise_ref(_unet(), c["noise"], c["cond"], c["uncond"], guidance_scale=c["guidance"], timesteps=c["steps"], collect=lat)
    torch.save({"latents": {k: lat[k - 1] for k in c["checkpoints"]}}, os.path.join(OUT, "full_loop.pt"))


def pipeline():
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from oracle.clip_ref import CLIPTextModelRef, CLIPVisionModelRef
    from oracle.infer_ref import conditioning_ref, denoise_ref, draw_noise_ref
    from oracle.vae_ref import AutoencoderKLDecoderRef
    c = fs.pipeline_case()
    with fs.no_init():
        vis, txt = CLIPVisionModelRef().eval(), CLIPTextModelRef().eval()
        ia, ta = PhotoVerseAdapterRef(1024, 768, 5).eval(), PhotoVerseAdapterRef(1024, 768, 5).eval()
        vae = AutoencoderKLDecoderRef(with_encoder=True).eval()
    vis.load_state_dict(fs.vision_state()); txt.load_state_dict(fs.text_state())
    ia.load_state_dict(fs.adapter_state(3)); ta.load_state_dict(fs.adapter_state(4)); vae.load_state_dict(fs.vae_state())
    with torch.no_grad():
        feats = vis(c["example"]["pixel_values_clip"], output_hidden_states=True)
        cond, uncond = conditioning_ref(c["example"], vis, txt, ta, ia, c["layers"], token_index=c["token_index"], uncond_input_ids=c["uncond_ids"])
        lat = []
        denoise_ref(_unet(), draw_noise_ref(1, 4, 64, seed=c["noise_seed"]), cond, uncond, guidance_scale=c["guidance"], timesteps=c["steps"], collect=lat)
        img = vae.decode(lat[-1] / 0.18215).sample.clamp(-1, 1)
    torch.save({"clip_last": feats[0].half(), "clip_hidden_rows": {i: feats[2][i][:, ::16].clone() for i in (0, 4, 8, 12, 16, 20, 24)},
                "clip_pooled": feats[1], "text": cond[0], "ip": cond[1], "utext": uncond[0], "uip": uncond[1],
                "latents": {k: lat[k - 1] for k in c["checkpoints"]}, "image_f16": img.half(),
                "clamped_fraction": (img.abs() >= 1).float().mean().item()}, os.path.join(OUT, "full_pipeline.pt"))


def vae():
    from oracle.vae_ref import AutoencoderKLDecoderRef
    c = fs.vae_case()
    with fs.no_init():
        ref = AutoencoderKLDecoderRef(with_encoder=True).eval()
    ref.load_state_dict(fs.vae_state())
    with torch.no_grad():
        img = ref.decode(c["z"]).sample
        post = ref.encode(c["x"]).latent_dist
    torch.save({"decode_f16_half_res": img[:, :, ::2, ::2].half().clone(), "decode_norm": img.norm().item(), "mean": post.mean, "logvar": post.logvar},
               os.path.join(OUT, "full_vae.pt"))


def train():
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from oracle.clip_ref import CLIPTextModelRef
    from oracle.lora_ref import inject_adapter_in_model_ref
    from oracle.unet_ref import (Transformer2DModelRef, UNet2DConditionModelRef, get_visual_cross_attention_values_norm_ref,
                                 set_visual_cross_attention_adapter_ref)
    c = fs.train_case()
    with fs.no_init():
        r_unet = UNet2DConditionModelRef().eval()
        set_visual_cross_attention_adapter_ref(r_unet, (c["E"],))
        r_txt = CLIPTextModelRef().eval()
        r_ia, r_ta = PhotoVerseAdapterRef(1024, 768, c["E"]).eval(), PhotoVerseAdapterRef(1024, 768, c["E"]).eval()
    r_unet.load_state_dict(fs.unet_state())
    inject_adapter_in_model_ref(r_unet, **fs.TRAIN_LORA)      # peft's un-merged forward: W x + (alpha / r) B A x (dropout 0)
    fs.fill_lora_(r_unet)
    r_txt.load_state_dict(fs.text_state()); r_ia.load_state_dict(fs.adapter_state(3)); r_ta.load_state_dict(fs.adapter_state(4))
    for p in list(r_unet.parameters()) + list(r_txt.parameters()):
        p.requires_grad_(False)
    r_params = dict(r_unet.named_parameters())
    train_names = [n for n in r_params if "to_k_ip" in n or "to_v_ip" in n or "lora_" in n]
    for n in train_names:
        r_params[n].requires_grad_(True)
    for name, m in r_unet.named_modules():
        if isinstance(m, Transformer2DModelRef):
            m.transformer_blocks[0].attn2.processor.forced_fusion_seed = c["forced"].get(name, c["forced"]["default"])
    e32 = [e.float() for e in c["embs"]]
    concept = r_ta(e32)
    ehs = r_txt({"text_input_ids": c["ids"], "concept_text_embeddings": concept, "concept_placeholder_idx": c["pidx"]})[0]
    ehs_img = r_ia(e32)
    with torch.enable_grad():
        pred = r_unet(c["noisy"], c["timesteps"], encoder_hidden_states=(ehs, ehs_img)).sample
        vn = get_visual_cross_attention_values_norm_ref(r_unet)
        loss = F.mse_loss(pred, c["noise"]) + 0.01 * concept.abs().mean() + 0.001 * vn.mean()
        loss.backward()

    def pack(named):
        return {n: {"norm": (p.grad if p.grad is not None else torch.zeros_like(p)).norm().item(),
                    "sub": fs.subsample(p.grad if p.grad is not None else torch.zeros_like(p))} for n, p in named}
    out = {"loss": loss.item(), "pred": pred.detach(), "unet": pack([(n, r_params[n]) for n in train_names]),
           "image_adapter": pack(r_ia.named_parameters()), "text_adapter": pack(r_ta.named_parameters())}
    # the same step WITHOUT the 0.01 * mean|concept| term (loss_weights = (1, 0, 0.001)): |x| has a kink at 0, and a concept element whose
    # sign differs between the fp16 device path and this fp32 run flips a whole +-0.01 / N contribution of the text adapter's gradient
    # (a few of the 3840 elements always sit inside the fp16 noise band) - the smooth variant pins the text-adapter chain tightly
    for m in (r_unet, r_ia, r_ta):
        m.zero_grad(set_to_none=True)
    concept = r_ta(e32)
    ehs = r_txt({"text_input_ids": c["ids"], "concept_text_embeddings": concept, "concept_placeholder_idx": c["pidx"]})[0]
    ehs_img = r_ia(e32)
    with torch.enable_grad():
        pred = r_unet(c["noisy"], c["timesteps"], encoder_hidden_states=(ehs, ehs_img)).sample
        vn = get_visual_cross_attention_values_norm_ref(r_unet)
        loss2 = F.mse_loss(pred, c["noise"]) + 0.001 * vn.mean()
        loss2.backward()
    out.update(loss_smooth=loss2.item(), text_adapter_smooth=pack(r_ta.named_parameters()), concept=concept.detach())
    torch.save(out, os.path.join(OUT, "full_train.pt"))


def tiny50():
    """The tiny-config 50-step loop of tests/test_unet_gpu.py::test_fifty_step_loop_latent_tolerance (30 s of oracle time on the GPU box)."""
    from oracle.infer_ref import denoise_ref, draw_noise_ref
    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
    torch.manual_seed(0)
    ref = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_visual_cross_attention_adapter_ref(ref, (5,))
    g = torch.Generator().manual_seed(31)
    cond = (torch.randn(1, 77, 768, generator=g), torch.randn(1, 1, 768, generator=g))
    uncond = (torch.randn(1, 77, 768, generator=g), torch.randn(1, 1, 768, generator=g))
    lat = denoise_ref(ref, draw_noise_ref(1, 4, 16, seed=6), cond, uncond, guidance_scale=7.5, timesteps=50)
    torch.save({"weights_seed": 0, "cond_seed": 31, "noise_seed": 6, "latents_50step": lat}, os.path.join(OUT, "full_tiny50.pt"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    todo = sys.argv[1:] or ["unet", "loop", "pipeline", "vae", "train", "tiny50"]
    for name in todo:
        t0 = time.time()
        globals()[name]()
        print(f"wrote full_{name}.pt in {time.time() - t0:.0f} s", flush=True)
