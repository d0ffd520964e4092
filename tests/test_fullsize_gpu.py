"""GPU parity at the FULL model sizes against expectations computed by the fp32 oracle IN THE BUILD CONTAINER
(``oracle/make_fullsize_golden.py`` -> ``tests/golden/full_*.pt``).  The GPU box only rebuilds the seeded weights / inputs
(``oracle/fullsize.py``) and runs the HIP path; one test keeps the live oracle as a canary and proves that the box's oracle reproduces the
committed expectation.  (Round 2 ran the oracle live in every one of these tests: ~200 s of host time per suite run.)"""
import os

import pytest
import torch

from oracle import fullsize as fs

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=True)


# fp16-storage tolerance for ONE full-size UNet forward vs the fp32 oracle (measured 1.0-1.2e-3)
TOL_FWD = 2.5e-3


def test_full_sd15_unet_forward_matches_live_oracle_and_fixture(full_hip_unet, full_weights, golden_dir):
    """configs[0]-shaped check at the REAL model size, bs=1, 64x64 latent.  CANARY: the fp32 oracle also runs live here, and must reproduce
    the committed expectation - which ties every other fixture of this file to an oracle this box can run."""
    from oracle.unet_ref import UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
    c, exp = fs.forward_case(), _load(golden_dir, "full_unet.pt")["forward_eps"]
    with fs.no_init():
        ref = UNet2DConditionModelRef().eval()
        set_visual_cross_attention_adapter_ref(ref, (5,))
    ref.load_state_dict(full_weights("unet"))
    assert sum(p.numel() for n, p in ref.named_parameters() if "processor" not in n) == 859_520_964     # public SD-v1.5 UNet size
    with torch.no_grad():
        live = ref(c["x"], torch.tensor(c["t"]), encoder_hidden_states=(c["text"], c["ip"])).sample
        got = full_hip_unet(c["x"].cuda(), torch.tensor(c["t"]), encoder_hidden_states=(c["text"].cuda(), c["ip"].cuda())).sample
    print(f"full SD-v1.5 UNet forward: live oracle vs fixture {rel_l2(live, exp):.2e}; HIP vs fixture {rel_l2(got, exp):.3e}")
    assert rel_l2(live, exp) < 2e-5             # same arithmetic on another host (thread count changes the fp32 summation order only)
    assert rel_l2(got, exp) < TOL_FWD
    del ref


def test_cfg4_per_rank_shape_forward(full_hip_unet, golden_dir):
    """BASELINE configs[4] per-rank shape: B=4, 96x96 latents (768x768), P=6 image tokens: N=9216 self-attention, the 12x12 level whose
    144 pixels are not a multiple of 64 (GroupNorm statistics fall back to the stats pass), 6 image-token K/V rows."""
    c, exp = fs.cfg4_case(), _load(golden_dir, "full_unet.pt")["cfg4_eps"]
    with torch.no_grad():
        got = full_hip_unet(c["x"].cuda(), torch.tensor(c["t"]), encoder_hidden_states=(c["text"].cuda(), c["ip"].cuda())).sample.cpu()
    for i in c["samples"]:
        err = rel_l2(got[i:i + 1], exp[i])
        print(f"cfg4 shape (B=4, 96x96, P=6) sample {i}: rel-L2 vs fp32 oracle = {err:.3e}")
        assert err < TOL_FWD


def test_headline_schedule_latents_within_north_star_tolerance(full_hip_unet, golden_dir):
    """The north_star number - latents within 1e-3 rel-L2 of the fp32 reference path - at the FULL model size, B=1, 64x64 latents,
    guidance 7.5, on the HEADLINE 50-step DPM-Solver++ schedule: after 10 steps and after ALL 50 (100 UNet forwards on each side; the
    oracle's 5 minutes ran in the build container).  The error is fp16 activation-storage noise (profiles/r02_fp16_noise_budget.txt)."""
    from photoverse_amd.pipeline import DenoiseLoop
    c, exp = fs.loop_case(), _load(golden_dir, "full_loop.pt")["latents"]
    loop = DenoiseLoop(full_hip_unet, 1, 64, 1, c["steps"], c["guidance"])
    loop.set_conditioning(tuple(t.cuda() for t in c["cond"]), tuple(t.cuda() for t in c["uncond"]))
    loop.reset(c["noise"])
    got10 = loop.run(10).clone().cpu()
    got50 = loop.run(40).clone().cpu()
    assert loop.state[0].item() == 50
    e10, e50 = rel_l2(got10, exp[10]), rel_l2(got50, exp[50])
    print(f"full-size latents on the 50-step schedule: after 10 steps {e10:.3e}, after 50 steps {e50:.3e} (rel-L2 vs fp32 oracle)")
    assert e10 < 1e-3 and e50 < 1e-3
    del loop


def test_headline_batch16_plan_latents_within_north_star_tolerance(full_hip_unet, golden_dir):
    """The same 1e-3 bound on the plan ``bench.py`` times: batch 16, the DEFAULT launch plans (two-stream heads / tails, merged low-resolution plan with its
    split-K heuristics, 8-wave self-attention, LDS-resident-patch convs, fused attn2), constructed with ``bench.py``'s own arguments.  The oracle's B = 1
    expectation is a batch member: the fixture's noise and conditioning sit at batch position 5, fifteen other seeded samples around it (samples never
    interact inside the UNet: ``test_unet_gpu.py`` proves that bit for bit with split-K off; here the headline plan itself meets the fp32 oracle)."""
    from photoverse_amd.pipeline import DenoiseLoop
    c, exp = fs.loop_case(), _load(golden_dir, "full_loop.pt")["latents"]
    B, pos = 16, 5
    g = torch.Generator().manual_seed(77)
    fill = lambda ref: torch.cat([torch.randn(B, *ref.shape[1:], generator=g)[:pos], ref, torch.randn(B, *ref.shape[1:], generator=g)[:B - pos - 1]])
    cond, uncond, noise = tuple(fill(t) for t in c["cond"]), tuple(fill(t) for t in c["uncond"]), fill(c["noise"])
    loop = DenoiseLoop(full_hip_unet, B, 64, 1, c["steps"], c["guidance"], use_graph=True, two_streams=True, batch_splits=1, share_prefix=False)
    loop.set_conditioning(tuple(t.cuda() for t in cond), tuple(t.cuda() for t in uncond))
    loop.reset(noise)
    got10 = loop.run(10)[pos:pos + 1].clone().cpu()
    got50 = loop.run(40)[pos:pos + 1].clone().cpu()
    assert loop.state[0].item() == 50 and torch.isfinite(loop.latents).all()
    e10, e50 = rel_l2(got10, exp[10]), rel_l2(got50, exp[50])
    print(f"headline plan (bs = 16) on the 50-step schedule, the oracle's sample at batch position {pos}: after 10 steps {e10:.3e}, after 50 steps {e50:.3e}")
    assert e10 < 1e-3 and e50 < 1e-3
    del loop


def _pipeline_models(full_hip_unet, full_weights):
    from photoverse_amd.adapters import PhotoVerseAdapter
    from photoverse_amd.clip import CLIPTextModel, CLIPVisionModel, patch_clip_text_transformer
    from photoverse_amd.vae import AutoencoderKL
    with fs.no_init():
        vis, txt = CLIPVisionModel(), patch_clip_text_transformer(CLIPTextModel())
        ia, ta = PhotoVerseAdapter(1024, 768, 5), PhotoVerseAdapter(1024, 768, 5)
        vae = AutoencoderKL()
    for m, key in ((vis, "vision"), (txt, "text"), (ia, "image_adapter"), (ta, "text_adapter"), (vae, "vae")):
        m.load_state_dict(full_weights(key))
        m.to("cuda")
    return vis, txt, ia, ta, vae


def test_clip_vit_l14_full_size(full_weights, golden_dir):
    """CLIP ViT-L/14 at its real size (24 layers, 257 tokens, 303 M parameters; infer.py:76-78): last hidden state and the hidden states
    the adapters consume (infer.py:80-84), vs the fp32 oracle (itself pinned against the installed transformers model, test_oracle_pins)."""
    from photoverse_amd.clip import CLIPVisionModel
    exp = _load(golden_dir, "full_pipeline.pt")
    with fs.no_init():
        vis = CLIPVisionModel()
    vis.load_state_dict(full_weights("vision"))
    vis.to("cuda")
    with torch.no_grad():
        got = vis(fs.pipeline_case()["example"]["pixel_values_clip"].cuda(), output_hidden_states=True)
    assert got[0].shape == (1, 257, 1024) and len(got[2]) == 25
    errs = {i: rel_l2(got[2][i][:, ::16], h) for i, h in exp["clip_hidden_rows"].items()}
    e_last = rel_l2(got[0], exp["clip_last"].float())
    print("full-size CLIP ViT-L/14: last hidden state", f"{e_last:.3e}", "hidden states", {k: f"{v:.2e}" for k, v in errs.items()})
    assert e_last < 3e-3 and max(errs.values()) < 3e-3


def test_whole_generation_full_size(full_hip_unet, full_weights, golden_dir):
    """``run_inference`` end to end at the real sizes (infer.py:72-123): CLIP ViT-L/14 -> both adapters (token_index 0) -> injected CLIP
    text encoder -> 50-step CFG loop (guidance 7.5) -> VAE decode + clamp, vs the fp32 oracle composition of the same steps."""
    from types import SimpleNamespace

    from photoverse_amd.infer import run_inference
    from photoverse_amd.scheduler import DPMSolverMultistepScheduler
    from photoverse_amd.tokenizer import load_tokenizer
    c, exp = fs.pipeline_case(), _load(golden_dir, "full_pipeline.pt")
    vis, txt, ia, ta, vae = _pipeline_models(full_hip_unet, full_weights)
    tok = load_tokenizer(None)
    assert tok([""], padding="max_length", max_length=77, return_tensors="pt").input_ids.tolist() == c["uncond_ids"].tolist()
    scheduler = SimpleNamespace(config=DPMSolverMultistepScheduler().config)
    with torch.no_grad():
        lat = run_inference(c["example"], tok, vis, txt, full_hip_unet, ta, ia, None, scheduler, "cuda", c["layers"], latent_size=64,
                            guidance_scale=c["guidance"], timesteps=c["steps"], token_index=c["token_index"], seed=c["noise_seed"]).cpu()
        img = run_inference(c["example"], tok, vis, txt, full_hip_unet, ta, ia, vae, scheduler, "cuda", c["layers"], latent_size=64,
                            guidance_scale=c["guidance"], timesteps=c["steps"], token_index=c["token_index"], seed=c["noise_seed"]).cpu()
    e_lat, e_img = rel_l2(lat, exp["latents"][50]), rel_l2(img, exp["image_f16"].float())
    print(f"whole generation at full size: final latents {e_lat:.3e}, 512x512 images {e_img:.3e} (clamped fraction {exp['clamped_fraction']:.3f})")
    assert img.shape == (1, 3, 512, 512) and float(img.abs().max()) <= 1.0
    assert e_lat < 3e-3 and e_img < 5e-3


def test_full_size_vae_decode_and_encode(full_weights, golden_dir):
    """SD-v1.5 VAE at its real size (83.65 M parameters): 64x64 latent -> 512x512 image, 256x256 image -> 32x32 posterior."""
    from photoverse_amd.vae import AutoencoderKL
    c, exp = fs.vae_case(), _load(golden_dir, "full_vae.pt")
    sd = full_weights("vae")
    assert sum(v.numel() for v in sd.values()) == 83_653_863          # public SD AutoencoderKL size
    with fs.no_init():
        hip = AutoencoderKL()
    hip.load_state_dict(sd)
    hip.to("cuda")
    with torch.no_grad():
        img = hip.decode(c["z"].cuda()).sample
        post = hip.encode(c["x"].cuda()).latent_dist
    e_dec = rel_l2(img[:, :, ::2, ::2], exp["decode_f16_half_res"].float())
    e_mean, e_lv = rel_l2(post.mean, exp["mean"]), rel_l2(post.logvar, exp["logvar"])
    print(f"full-size VAE: decode {e_dec:.3e} (norm {float(img.norm()):.2f} vs {exp['decode_norm']:.2f}), encode mean {e_mean:.3e} logvar {e_lv:.3e}")
    assert img.shape == (1, 3, 512, 512) and e_dec < 5e-3 and float(img.norm()) == pytest.approx(exp["decode_norm"], rel=2e-3)
    assert post.mean.shape == (1, 4, 32, 32) and e_mean < 5e-3 and e_lv < 5e-3


@pytest.mark.parametrize("attn_bwd_min", ["128", "1"])
def test_full_size_training_gradients(full_weights, golden_dir, monkeypatch, attn_bwd_min):
    """``attn_bwd_min`` = PV_ATTN8_BWD_MIN: "128" is the library's rule (at B = 1 the attention backward runs on the 4-wave passes), "1" puts the 8-wave
    staggered passes of pv_attnbwd.hip (d = 40 and d = 80) into the same plan - the form the bs = 16 training step uses.
    The training backward at the FULL model sizes (SD-v1.5 UNet, 12-layer CLIP text encoder, 1024-wide adapters, 5 tokens, LoRA r=8),
    B=1, 64x64 latents: gradients of every trainable group against torch autograd over the fp32 oracle WITH THE INDEPENDENT peft
    restatement (oracle/lora_ref.py: un-merged W x + (alpha / r) B A x) - computed in the build container; here: per-group rel-L2 on
    strided sub-samples of every gradient tensor and per-group norms."""
    from photoverse_amd.adapters import PhotoVerseAdapter
    from photoverse_amd.clip import CLIPTextModel
    from photoverse_amd.lora import LoraConfig, inject_adapter_in_model
    from photoverse_amd.train import TrainStep
    from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter
    monkeypatch.setenv("PV_ATTN8_BWD_MIN", attn_bwd_min)
    c, exp = fs.train_case(), _load(golden_dir, "full_train.pt")
    E = c["E"]
    with fs.no_init():
        unet = UNet2DConditionModel()
        set_visual_cross_attention_adapter(unet, (E,))
        text_encoder = CLIPTextModel()
        image_adapter, text_adapter = PhotoVerseAdapter(1024, 768, E), PhotoVerseAdapter(1024, 768, E)
    unet.load_state_dict(full_weights("unet"))
    inject_adapter_in_model(LoraConfig(r=fs.TRAIN_LORA["r"], lora_alpha=fs.TRAIN_LORA["lora_alpha"], target_modules=fs.TRAIN_LORA["target_modules"]), unet)
    fs.fill_lora_(unet)
    text_encoder.load_state_dict(full_weights("text"))
    image_adapter.load_state_dict(full_weights("image_adapter"))
    text_adapter.load_state_dict(full_weights("text_adapter"))
    for m in (unet, text_encoder, image_adapter, text_adapter):
        m.to("cuda")
    ts = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=1, h=64, w=64, n_tokens=E, grad_scale=4096.0, fusion_seed=3)
    forced = [c["forced"].get(name, c["forced"]["default"]) for name in ts.fusion_names]
    assert len(forced) == 16 and sorted(set(forced)) == [0.1, 0.5, 0.9]
    out = ts.step(noisy_latents=c["noisy"].cuda(), noise=c["noise"].cuda(), timesteps=c["timesteps"], text_input_ids=c["ids"].cuda(),
                  placeholder_idx=c["pidx"].cuda(), image_embeddings=[e.cuda() for e in c["embs"]], forced_fusion=forced)
    torch.cuda.synchronize()
    assert out["loss"].item() == pytest.approx(exp["loss"], rel=5e-3)
    S = ts.grad_scale
    h_params = dict(unet.named_parameters())
    assert set(exp["unet"]) == {n for n in h_params if "to_k_ip" in n or "to_v_ip" in n or "lora_" in n}     # same trainable names as peft / the reference

    def group(pairs):
        got = torch.cat([fs.subsample(hp.grad.float().cpu() / S) for hp, _ in pairs])
        want = torch.cat([e["sub"] for _, e in pairs])
        n_got = torch.stack([(hp.grad.float() / S).norm().cpu() for hp, _ in pairs]).norm().item()
        n_want = torch.tensor([e["norm"] for _, e in pairs]).norm().item()
        return rel_l2(got, want), n_got / n_want
    groups = dict(ip=[(h_params[n], e) for n, e in exp["unet"].items() if "_ip" in n],
                  lora_A=[(h_params[n], e) for n, e in exp["unet"].items() if "lora_A" in n],
                  lora_B=[(h_params[n], e) for n, e in exp["unet"].items() if "lora_B" in n],
                  image_adapter=[(p, exp["image_adapter"][n]) for n, p in image_adapter.named_parameters()],
                  text_adapter=[(p, exp["text_adapter"][n]) for n, p in text_adapter.named_parameters()])
    assert len(groups["ip"]) == 32 and len(groups["lora_A"]) == 48
    res = {k: group(v) for k, v in groups.items()}
    print("full-size training-step gradients (rel-L2 on sub-samples, norm ratio) per group:", {k: (f"{e:.2e}", f"{r:.4f}") for k, (e, r) in res.items()})
    # measured: ip 8.5e-4, LoRA A / B 8.8e-4 / 9.5e-4, image adapter 1.4e-3.  The text adapter additionally carries the 0.01 * mean|concept|
    # term, whose gradient is +-0.01 / N per concept element: one element of 3840 whose sign differs between the fp16 device path and
    # the fp32 oracle moves the group by ~2e-2 (measured 1.0e-2 .. 2.2e-2) - a kink of the loss, not of the kernels ...
    assert max(res["ip"][0], res["lora_A"][0], res["lora_B"][0]) < 3e-3, res
    # the kinked text-adapter value is REPORTED (1.0e-2 .. 2.2e-2 over the boxes seen) with a sanity bound only; the asserted bound for that
    # chain is the smooth variant's below
    print(f"text adapter WITH the |concept| kink term: rel-L2 {res['text_adapter'][0]:.3e} (reported; the smooth variant below is the asserted one)")
    assert res["image_adapter"][0] < 4e-3 and res["text_adapter"][0] < 3.5e-2, res
    assert all(abs(r - 1.0) < 5e-3 for _, r in res.values()), res
    # ... so the text-adapter chain (adapter MLPs <- injected 12-layer text encoder <- K / V of 16 cross-attention layers) is pinned on the
    # same step without that term (loss_weights = (1, 0, 0.001)), where it is smooth
    for pr in list(text_adapter.parameters()):
        pr.grad = None
    ts2 = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=1, h=64, w=64, n_tokens=E, grad_scale=4096.0, fusion_seed=3,
                    loss_weights=(1.0, 0.0, 0.001))
    out2 = ts2.step(noisy_latents=c["noisy"].cuda(), noise=c["noise"].cuda(), timesteps=c["timesteps"], text_input_ids=c["ids"].cuda(),
                    placeholder_idx=c["pidx"].cuda(), image_embeddings=[e.cuda() for e in c["embs"]], forced_fusion=forced)
    torch.cuda.synchronize()
    assert out2["loss"].item() == pytest.approx(exp["loss_smooth"], rel=5e-3)
    e_smooth, r_smooth = group([(p_, exp["text_adapter_smooth"][n]) for n, p_ in text_adapter.named_parameters()])
    print(f"text adapter without the |concept| term: rel-L2 {e_smooth:.3e}, norm ratio {r_smooth:.4f}")
    # measured 8.0e-3 (fp16 gradients through the 12-layer text encoder and the adapter's LeakyReLU / LayerNorm kinks; norm ratio 1.0003)
    assert e_smooth < 1.2e-2 and abs(r_smooth - 1.0) < 2e-3
