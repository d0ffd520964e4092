"""GPU parity of the pre-loop conditioning stack (infer.py:76-96) and of run_inference end to end vs the CPU oracle."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

VIS = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=3, image_size=56, patch_size=14)
TXT = dict(vocab_size=49408, hidden_size=768, num_attention_heads=12, intermediate_size=512, num_hidden_layers=2, max_position_embeddings=77)


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.fixture(scope="module")
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")


def test_clip_vision_matches_oracle(need_gpu):
    from oracle.clip_ref import CLIPVisionModelRef
    from photoverse_amd.clip import CLIPVisionModel
    torch.manual_seed(0)
    ref = CLIPVisionModelRef(**VIS).eval()
    hip = CLIPVisionModel(**VIS)
    hip.load_state_dict(ref.state_dict())
    hip.to("cuda")
    x = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        exp = ref(x)
        got = hip(x.cuda(), output_hidden_states=True)
    assert len(got[2]) == len(exp[2]) == 4 and got[0].shape == (2, 17, 256)
    for i in range(4):
        assert rel_l2(got[2][i], exp[2][i]) < 3e-3
    assert rel_l2(got[0], exp[0]) < 3e-3 and rel_l2(got[1], exp[1]) < 3e-3
    assert torch.equal(got[0], got.hidden_states[-1])


def test_clip_text_with_injection_matches_oracle(need_gpu):
    from oracle.clip_ref import CLIPTextModelRef
    from photoverse_amd.clip import CLIPTextModel, patch_clip_text_transformer
    torch.manual_seed(0)
    ref = CLIPTextModelRef(**TXT).eval()
    hip = patch_clip_text_transformer(CLIPTextModel(**TXT))
    hip.load_state_dict(ref.state_dict())
    hip.to("cuda")
    g = torch.Generator().manual_seed(2)
    ids = torch.randint(0, 1000, (3, 77), generator=g)
    with torch.no_grad():
        exp = ref({"text_input_ids": ids})[0]
        got = hip({"text_input_ids": ids.cuda()})[0]
        assert rel_l2(got, exp) < 3e-3
        for E, idx in ((1, torch.tensor([[5], [1], [71]])), (5, torch.tensor([5, 1, 71]))):
            concept = torch.randn(3, E, 768, generator=g)
            exp = ref({"text_input_ids": ids, "concept_text_embeddings": concept, "concept_placeholder_idx": idx})[0]
            got = hip({"text_input_ids": ids.cuda(), "concept_text_embeddings": concept.cuda(), "concept_placeholder_idx": idx.cuda()})[0]
            assert rel_l2(got, exp) < 3e-3
    with pytest.raises(ValueError):
        hip(None)
    with pytest.raises(IndexError):
        hip({"text_input_ids": torch.full((1, 77), 49408).cuda()})          # out-of-vocabulary id fails loudly
    with pytest.raises(TypeError):
        patch_clip_text_transformer(torch.nn.Linear(2, 2))


def test_adapter_matches_real_reference_golden(need_gpu, golden_dir):
    """HIP adapter vs vectors produced by the REAL /root/reference/models/adapters.py (oracle/make_golden.py)."""
    from photoverse_amd.adapters import PhotoVerseAdapter
    g = torch.load(os.path.join(golden_dir, "adapter_golden.pt"))
    torch.manual_seed(g["weights_seed"])
    ad = PhotoVerseAdapter(1024, 768, num_tokens=2)          # same registration order -> same default init as the reference
    for k, (s1, _, head) in g["weight_checksums"].items():
        assert torch.equal(ad.state_dict()[k].flatten()[:4], head)
    ad.to("cuda")
    embs = [e.cuda() for e in g["embs"]]
    for key, ti in (("none", None), ("full", "full"), ("0", 0), ("1", 1)):
        out = ad(embs, token_index=ti)
        assert out.shape == g["outs"][key].shape
        assert rel_l2(out, g["outs"][key]) < 3e-3


def test_run_inference_end_to_end_matches_oracle(need_gpu):
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from oracle.clip_ref import CLIPTextModelRef, CLIPVisionModelRef
    from oracle.infer_ref import conditioning_ref, denoise_ref, draw_noise_ref
    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
    from photoverse_amd.infer import run_inference
    from photoverse_amd.modeling_utils import load_models
    tok, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = load_models(
        None, 1, unet_config=TINY_CONFIG, vision_config=VIS, text_config=TXT, seed=3)
    from photoverse_amd.vae import AutoencoderKL
    assert isinstance(vae, AutoencoderKL)
    # oracle twins with identical weights
    r_unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_visual_cross_attention_adapter_ref(r_unet, (2,))
    r_unet.load_state_dict(unet.state_dict())
    r_vis = CLIPVisionModelRef(**VIS).eval(); r_vis.load_state_dict(image_encoder.state_dict())
    r_txt = CLIPTextModelRef(**TXT).eval(); r_txt.load_state_dict(text_encoder.state_dict())
    r_ia = PhotoVerseAdapterRef(256, 768, 2).eval(); r_ia.load_state_dict(image_adapter.state_dict())
    r_ta = PhotoVerseAdapterRef(256, 768, 2).eval(); r_ta.load_state_dict(text_adapter.state_dict())
    for m in (unet, text_encoder, image_encoder, image_adapter, text_adapter):
        m.to("cuda")
    g = torch.Generator().manual_seed(4)
    B = 2
    example = {"pixel_values": torch.zeros(B, 3, 128, 128), "pixel_values_clip": torch.randn(B, 3, 56, 56, generator=g),
               "text_input_ids": torch.randint(0, 1000, (B, 77), generator=g), "concept_placeholder_idx": torch.tensor([[5], [3]])}
    for token_index, layers in ((0, [1]), ("full", [1])):
        with torch.no_grad():
            got = run_inference(example, tok, image_encoder, text_encoder, unet, text_adapter, image_adapter, None, scheduler, "cuda",
                                layers, latent_size=16, guidance_scale=3.0, timesteps=3, token_index=token_index, seed=9)
            uids = tok([""] * B, padding="max_length", max_length=77, return_tensors="pt").input_ids
            cond, uncond = conditioning_ref(example, r_vis, r_txt, r_ta, r_ia, layers, token_index=token_index, uncond_input_ids=uids)
            exp = denoise_ref(r_unet, draw_noise_ref(B, 4, 16, seed=9), cond, uncond, guidance_scale=3.0, timesteps=3)
        assert got.shape == exp.shape
        err = rel_l2(got, exp)
        print(f"run_inference end to end (tiny config, token_index={token_index}): latents rel-L2 vs oracle {err:.3e}")
        assert err < 3e-3           # measured 6.5e-4; the full-size counterpart (tests/test_fullsize_gpu.py::test_whole_generation_full_size) asserts 3e-3
    with pytest.raises(NotImplementedError):
        run_inference(example, tok, image_encoder, text_encoder, unet, text_adapter, image_adapter, None, scheduler, "cuda", [1],
                      latent_size=16, timesteps=2, from_noised_image=True)


def test_generate_cli_end_to_end(need_gpu, tmp_path):
    """Row G: the CLI counterpart of /root/reference/generate.py runs as a program - load_models -> preprocessing ->
    run_inference (CLIP, adapters, text encoder, graph loop, VAE decode) -> PNG files - once from synthetic pixels and once from
    an image file through the reference-exact preprocessing (short-side resize + centre crop)."""
    import subprocess
    import sys
    import numpy as np
    from PIL import Image
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "generate.py"), "--model_path", "random", "--tiny", "--num_timesteps", "2", "--latent_size", "16",
            "--num_of_samples", "2", "--seed", "3", "--guidance_scale", "2.0", "--encoder_layers_idx", "1", "2"]
    out1 = tmp_path / "synthetic"
    r = subprocess.run(base + ["--synthetic_input", "--results_dir", str(out1)], capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    files = sorted(os.listdir(out1))
    assert files == ["generated_image0.png", "generated_image1.png"]
    a = np.asarray(Image.open(out1 / files[0]))
    assert a.shape == (128, 128, 3) and a.dtype == np.uint8 and a.std() > 0
    rng = np.random.default_rng(0)
    face = tmp_path / "face.png"
    Image.fromarray(rng.integers(0, 256, (300, 200, 3), dtype=np.uint8)).save(face)
    out2 = tmp_path / "from_file"
    r = subprocess.run(base + ["--input_image_path", str(face), "--results_dir", str(out2), "--from_noised_image", "--negative_prompt", "blurry"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    assert sorted(os.listdir(out2)) == files
    b = np.asarray(Image.open(out2 / files[1]))
    assert b.shape == (128, 128, 3) and b.std() > 0


def test_reference_layout_checkpoint_loads_into_hip_path(need_gpu, tmp_path):
    """A `photoverse_000001.pt` in the reference's layout (modeling_utils.py:29-50: image_adapter, text_adapter, the attn2
    to_q/to_k/to_v + processor subset WITH peft-style LoRA keys, lora_config) is loaded by load_models into a fresh HIP model; the
    UNet forward and both adapters then match the oracle loaded from the SAME file (LoRA applied as W + alpha/r B A)."""
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
    from photoverse_amd.lora import LoraConfig
    from photoverse_amd.modeling_utils import load_models, save_progress
    cfg = LoraConfig(r=4, lora_alpha=8)
    kw = dict(unet_config=TINY_CONFIG, vision_config=VIS, text_config=TXT)
    # "trained" model: seeded init + non-zero LoRA B, written in the reference layout
    _t, _te, _v, unet_a, _ie, ia_a, ta_a, _s, _ = load_models(None, 1, use_lora=True, lora_config=cfg, seed=11, **kw)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for n, p in unet_a.named_parameters():
            if "lora_B" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    save_progress(ia_a, ta_a, unet_a, None, str(tmp_path), step=1, lora_config=cfg)
    ck_path = str(tmp_path / "photoverse_000001.pt")
    ck = torch.load(ck_path)
    assert any("lora_A.default.weight" in k for k in ck["cross_attention_adapter"]) and ck["lora_config"]["r"] == 4
    # fresh model (same base seed = the frozen SD weights, which the checkpoint does not carry) + the checkpoint, on the device
    _t, _te, _v, unet_b, _ie, ia_b, ta_b, _s, lc = load_models(None, 1, ck_path, seed=11, **kw)
    assert lc is not None and lc.r == 4
    for m in (unet_b, ia_b, ta_b):
        m.to("cuda")
    # oracle from the same file: base weights of the same seed, cross-attention subset from the file with LoRA merged by hand
    ref = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_visual_cross_attention_adapter_ref(ref, (2,))
    base_sd = {k: v for k, v in unet_a.state_dict().items() if "lora_" not in k}
    base_sd = {k.replace(".base_layer.", "."): v for k, v in base_sd.items()}
    ca = ck["cross_attention_adapter"]
    for k in list(ca):
        if k.endswith("base_layer.weight"):
            stem = k[: -len("base_layer.weight")]
            base_sd[stem + "weight"] = ca[k] + (cfg.lora_alpha / cfg.r) * (ca[stem + "lora_B.default.weight"] @ ca[stem + "lora_A.default.weight"])
        elif "lora_" not in k:
            base_sd[k] = ca[k]
    ref.load_state_dict(base_sd)
    x, text, ip = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 77, 768, generator=g), torch.randn(2, 2, 768, generator=g)
    with torch.no_grad():
        exp = ref(x, torch.tensor(400), encoder_hidden_states=(text, ip)).sample
        got = unet_b(x.cuda(), torch.tensor(400), encoder_hidden_states=(text.cuda(), ip.cuda())).sample
    assert rel_l2(got, exp) < 2.5e-3
    # without the LoRA term the result must differ measurably (the test would otherwise not see a dropped adapter)
    for k in list(ca):
        if k.endswith("base_layer.weight"):
            base_sd[k[: -len("base_layer.weight")] + "weight"] = ca[k]
    ref.load_state_dict(base_sd)
    with torch.no_grad():
        assert rel_l2(got, ref(x, torch.tensor(400), encoder_hidden_states=(text, ip)).sample) > 3 * rel_l2(got, exp)
    r_ia = PhotoVerseAdapterRef(256, 768, 2).eval(); r_ia.load_state_dict(ck["image_adapter"])
    r_ta = PhotoVerseAdapterRef(256, 768, 2).eval(); r_ta.load_state_dict(ck["text_adapter"])
    embs = [torch.randn(2, 17, 256, generator=g) for _ in range(2)]
    with torch.no_grad():
        assert rel_l2(ia_b([e.cuda() for e in embs]), r_ia(embs)) < 3e-3
        assert rel_l2(ta_b([e.cuda() for e in embs], token_index=0), r_ta(embs, token_index=0)) < 3e-3


def test_training_step_forward_losses_match_oracle(need_gpu):
    """Forward half of a training step (train.py:466-516): VAE encode + posterior sample, add_noise at per-sample timesteps, CLIP ->
    adapters in FULL mode (3 tokens), injected text encoder, UNet in grad mode (forced per-layer fusion draws), and the three loss
    terms - against the same composition of the oracle pieces with the same draws."""
    import torch.nn.functional as F
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from oracle.clip_ref import CLIPTextModelRef, CLIPVisionModelRef
    from oracle.scheduler_ref import DPMSolverMultistepRef
    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef, get_visual_cross_attention_values_norm_ref, set_visual_cross_attention_adapter_ref
    from oracle.vae_ref import AutoencoderKLDecoderRef
    from photoverse_amd.modeling_utils import load_models
    from photoverse_amd.train import training_step_forward
    VAE = dict(block_out_channels=(128, 128, 128, 128), layers_per_block=1)
    ENT = 2                                                    # extra_num_tokens -> P = 3
    tok, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = load_models(
        None, ENT, unet_config=TINY_CONFIG, vision_config=VIS, text_config=TXT, vae_config=VAE, seed=21)
    r_unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_visual_cross_attention_adapter_ref(r_unet, (ENT + 1,))
    r_unet.load_state_dict(unet.state_dict())
    r_vis = CLIPVisionModelRef(**VIS).eval(); r_vis.load_state_dict(image_encoder.state_dict())
    r_txt = CLIPTextModelRef(**TXT).eval(); r_txt.load_state_dict(text_encoder.state_dict())
    r_ia = PhotoVerseAdapterRef(256, 768, ENT + 1).eval(); r_ia.load_state_dict(image_adapter.state_dict())
    r_ta = PhotoVerseAdapterRef(256, 768, ENT + 1).eval(); r_ta.load_state_dict(text_adapter.state_dict())
    r_vae = AutoencoderKLDecoderRef(with_encoder=True, **VAE).eval(); r_vae.load_state_dict(vae.state_dict())
    for m in (unet, vae, text_encoder, image_encoder, image_adapter, text_adapter):
        m.to("cuda")
    g = torch.Generator().manual_seed(22)
    B = 2
    batch = {"pixel_values": torch.rand(B, 3, 128, 128, generator=g) * 2 - 1, "pixel_values_clip": torch.randn(B, 3, 56, 56, generator=g),
             "text_input_ids": torch.randint(0, 1000, (B, 77), generator=g), "concept_placeholder_idx": torch.tensor([[5], [3]])}
    noise = torch.randn(B, 4, 16, 16, generator=g)
    eps = torch.randn(B, 4, 16, 16, generator=g)
    timesteps = torch.tensor([731, 42])
    layers = [1, 2]
    forced = [0.1, 0.5, 0.9, 0.4]
    out = training_step_forward(batch, tok, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae, scheduler, "cuda", layers, ENT,
                                fusion_seed=3, forced_fusion=forced, noise=noise, timesteps=timesteps, posterior_eps=eps)
    assert out["fusion_table"].cpu().tolist() == [[2.0, 0.0], [1.0, 1.0], [0.0, 2.0], [1.0, 1.0]]
    # ---- oracle composition (train.py:466-516) ----
    with torch.no_grad():
        lat = r_vae.encode(batch["pixel_values"]).latent_dist.sample(eps=eps) * 0.18215
        sch = DPMSolverMultistepRef()
        acp = torch.as_tensor(sch.alphas_cumprod)[timesteps].float().view(-1, 1, 1, 1)
        noisy = acp.sqrt() * lat + (1 - acp).sqrt() * noise
        feats = r_vis(batch["pixel_values_clip"])
        embs = [feats[0]] + [feats[2][i] for i in layers]
        concept = r_ta(embs)
        ehs = r_txt({"text_input_ids": batch["text_input_ids"], "concept_text_embeddings": concept,
                     "concept_placeholder_idx": batch["concept_placeholder_idx"]})[0]
        ehs_img = r_ia(embs)
    mods = dict(r_unet.named_modules())
    for name, u in zip(out["fusion_names"], forced):
        mods[name + ".transformer_blocks.0.attn2"].processor.forced_fusion_seed = u
    with torch.enable_grad():
        pred = r_unet(noisy, timesteps, encoder_hidden_states=(ehs, ehs_img)).sample.detach()
        vn = get_visual_cross_attention_values_norm_ref(r_unet).detach()
    exp_diff = F.mse_loss(pred, noise).item()
    exp_concept = concept.abs().mean().item()
    exp_visual = vn.mean().item()
    assert rel_l2(out["latents"], lat) < 5e-3 and rel_l2(out["noise_pred"], pred) < 1e-2
    assert out["diffusion_loss"].item() == pytest.approx(exp_diff, rel=5e-3)
    assert out["concept_text_loss"].item() == pytest.approx(exp_concept, rel=3e-3)
    assert out["cross_attn_visual_loss"].item() == pytest.approx(exp_visual, rel=3e-3)
    assert out["loss"].item() == pytest.approx(exp_diff + 0.01 * exp_concept + 0.001 * exp_visual, rel=5e-3)
    # free draws: runs, finite, timesteps in range
    out2 = training_step_forward(batch, tok, image_encoder, text_encoder, unet, text_adapter, image_adapter, vae, scheduler, "cuda", layers, ENT,
                                 generator=torch.Generator().manual_seed(5), fusion_seed=3)
    assert torch.isfinite(out2["loss"]).all() and 0 <= int(out2["timesteps"].min()) and int(out2["timesteps"].max()) < 1000


def test_adapter_backward_matches_oracle_autograd(need_gpu):
    """BACKWARD of the reference's own adapters on HIP (what train.py:372-377 optimises): every parameter gradient of
    PhotoVerseAdapter - both mapping MLPs of every token, through LayerNorm + LeakyReLU and the patch-token mean - against torch
    autograd over the oracle adapter, in full mode and for a single token."""
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from photoverse_amd.adapters import PhotoVerseAdapter
    torch.manual_seed(5)
    ref = PhotoVerseAdapterRef(256, 768, 2)
    hip = PhotoVerseAdapter(256, 768, 2)
    hip.load_state_dict(ref.state_dict())
    hip.to("cuda")
    g = torch.Generator().manual_seed(6)
    B, T = 3, 17
    embs = [torch.randn(B, T, 256, generator=g) for _ in range(2)]
    for token_index, G in ((None, torch.randn(B, 2, 768, generator=g)), (1, torch.randn(B, 1, 768, generator=g))):
        for m in (ref, hip):
            m.zero_grad(set_to_none=True)
        with torch.enable_grad():
            out_r = ref(embs, token_index=token_index)
            (out_r * G).sum().backward()
            out_h = hip([e.cuda() for e in embs], token_index=token_index)
            assert out_h.requires_grad and rel_l2(out_h, out_r) < 3e-3
            (out_h * G.cuda()).sum().backward()
        gr = dict(ref.named_parameters())
        checked = 0
        for n, p in hip.named_parameters():
            if gr[n].grad is None:
                assert p.grad is None or p.grad.abs().max() == 0, n          # tokens the call did not use
                continue
            assert p.grad is not None and p.grad.shape == gr[n].grad.shape, n
            # the forward stores fp16 activations, so pre-activations within ~1e-3 of zero land on the other side of the LeakyReLU kink
            # than in the fp32 oracle (slope 1 vs 0.01): ~0.1 % of the units flip, which is a 2-3 % gradient difference after two layers
            assert rel_l2(p.grad, gr[n].grad) < 4e-2, (n, rel_l2(p.grad, gr[n].grad))
            checked += 1
        assert checked == (40 if token_index is None else 20)                 # per token: 2 MLPs x (3 Linear + 2 LayerNorm) x (weight, bias)


@pytest.mark.gpu
@pytest.mark.parametrize("grad_scale", [1.0, 256.0])
def test_adamw_and_clip_match_torch(grad_scale):
    """optim.AdamW (HIP kernels, device-side clip coefficient) against torch.optim.AdamW + clip_grad_norm_ on CPU fp32
    (train.py:372-377, :538-545): three steps, two clip groups + one unclipped parameter."""
    from photoverse_amd.optim import AdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(1024, 768), (1024,), (320, 768), (7,), (33, 5)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    o_ref = torch.optim.AdamW(ref, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    o_hip = AdamW(mine, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    groups = ([0, 1], [2, 3])
    for step in range(3):
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g) * (0.05 if step == 1 else 2.0)     # step 1: norm < 1 -> coefficient clamps at 1
            p.grad = gr.clone()
            q.grad = (gr * grad_scale).cuda()
        want_norms = [float(torch.nn.utils.clip_grad_norm_([ref[i] for i in grp], 1.0)) for grp in groups]
        o_ref.step()
        norms = o_hip.step(clip_groups=[[mine[i] for i in grp] for grp in groups], max_norm=1.0, grad_scale=grad_scale)
        for n, w in zip(norms, want_norms):
            assert abs(float(n) - w) < 1e-4 * w
        for p, q in zip(ref, mine):
            assert rel_l2(q.detach().cpu(), p.detach()) < 2e-6, step


@pytest.mark.gpu
def test_adamw_skips_a_step_whose_gradients_overflowed():
    """fp16 gradients under a static loss scale can overflow where the fp32 reference cannot: a non-finite group norm must SKIP the step
    (parameters, moments and the bias-correction count untouched - GradScaler semantics, decided on the device), not write NaN into
    every parameter of the group; the following steps equal torch.optim.AdamW stepping only on the finite gradients."""
    from photoverse_amd.optim import AdamW
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 48), (48,), (33, 5)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    o_ref = torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2)
    o_hip = AdamW(mine, lr=1e-2, weight_decay=1e-2)
    groups = ([0, 1], [2])
    for step, poison in enumerate((None, float("inf"), float("nan"), None, None)):
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g)
            p.grad = gr.clone()
            q.grad = (gr * 128.0).cuda()
        before = [q.detach().clone() for q in mine]
        if poison is not None:
            mine[2].grad[3, 1] = poison                    # one element of ONE group: the whole step is skipped, every group
        else:
            for grp in groups:
                torch.nn.utils.clip_grad_norm_([ref[i] for i in grp], 1.0)
            o_ref.step()
        o_hip.step(clip_groups=[[mine[i] for i in grp] for grp in groups], max_norm=1.0, grad_scale=128.0)
        for b, q, p in zip(before, mine, ref):
            assert torch.isfinite(q).all()
            if poison is not None:
                assert torch.equal(q.detach(), b)
            assert rel_l2(q.detach().cpu(), p.detach()) < 2e-6, step
    assert o_hip.skipped_steps == 2 and o_hip.applied_steps == 3 and o_hip.step_count == 5
    assert int(o_hip.state_dict()["state"][0]["step"]) == 3             # the checkpoint carries the applied count (torch's bias correction)


@pytest.mark.gpu
def test_adamw_guard_covers_ungrouped_tensors_and_a_poisoned_first_step():
    """A parameter in NO clip group must be skipped together with the grouped ones when a group overflows (it used to be updated, and with a
    skipped FIRST step its bias correction divided by 1 - beta^0 = 0); and a run that starts without clip groups and adds them later keeps
    its bias-correction count (moments carry N steps of history)."""
    from photoverse_amd.optim import AdamW
    g = torch.Generator().manual_seed(19)
    shapes = [(40, 24), (24,), (17, 3)]
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    o_ref = torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2)
    o_hip = AdamW(mine, lr=1e-2, weight_decay=1e-2)
    for step, poison in enumerate((float("inf"), None, float("nan"), None)):         # parameter 2 is in no group; the FIRST step is poisoned
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g)
            p.grad = gr.clone()
            q.grad = (gr * 64.0).cuda()
        before = [q.detach().clone() for q in mine]
        if poison is not None:
            mine[0].grad[5, 7] = poison
        else:
            torch.nn.utils.clip_grad_norm_([ref[0], ref[1]], 1.0)
            o_ref.step()
        o_hip.step(clip_groups=[[mine[0], mine[1]]], max_norm=1.0, grad_scale=64.0)
        for b, q, p in zip(before, mine, ref):
            assert torch.isfinite(q).all(), step
            if poison is not None:
                assert torch.equal(q.detach(), b), step                            # grouped AND ungrouped tensors untouched
            assert rel_l2(q.detach().cpu(), p.detach()) < 2e-6, step
    assert o_hip.skipped_steps == 2 and o_hip.applied_steps == 2
    # unguarded -> guarded transition: two plain steps, then a step with clip groups continues at step 3 of the bias correction
    ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    o_ref, o_hip = torch.optim.AdamW(ref, lr=1e-2, weight_decay=1e-2), AdamW(mine, lr=1e-2, weight_decay=1e-2)
    for step in range(4):
        for p, q in zip(ref, mine):
            gr = torch.randn(p.shape, generator=g)
            p.grad, q.grad = gr.clone(), gr.cuda()
        if step >= 2:
            torch.nn.utils.clip_grad_norm_([ref[0], ref[1]], 1.0)
        o_ref.step()
        o_hip.step(clip_groups=[[mine[0], mine[1]]] if step >= 2 else ())
        for q, p in zip(mine, ref):
            assert rel_l2(q.detach().cpu(), p.detach()) < 2e-6, step
    assert o_hip.applied_steps == 4


@pytest.mark.parametrize("p_drop,hw", [(0.0, 16), (0.1, 16), (0.0, 20)])
def test_training_step_backward_matches_oracle_autograd(need_gpu, p_drop, hw):
    """The whole backward of a training step (train.py:495-536 without the optional face loss) on the HIP plans: gradient of
    mse + 0.01 |concept| + 0.001 ||V_ip|| w.r.t. every trainable parameter - both adapters (through the UNet's cross-attention layers;
    the text adapter additionally through the CLIP text encoder), to_k_ip / to_v_ip of every processor and the LoRA factors behind
    attn2.to_q / to_k / to_v - against torch autograd over the fp32 oracle composition with the same forced fusion draws.  Then one
    AdamW step with the reference's per-module gradient clipping, against torch.optim.AdamW on the oracle.
    p_drop = 0.1: the reference's default lora_dropout (train.py:265) - the low-rank branches run un-merged with the device-side dropout;
    the oracle applies the SAME keep masks (rebuilt from the counter-based generator after the step).
    hw = 20: ragged everything - 400 / 100 tokens (attention tiles of 64 / 128 with tails, GroupNorm statistics without the 64-row column
    statistics, conv / GEMM row tails)."""
    import torch.nn.functional as F
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from oracle.clip_ref import CLIPTextModelRef
    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef, get_visual_cross_attention_values_norm_ref, set_visual_cross_attention_adapter_ref
    from photoverse_amd.lora import LoraConfig, LoRALinear, inject_adapter_in_model
    from photoverse_amd.modeling_utils import load_models
    from photoverse_amd.optim import AdamW
    from photoverse_amd.train import TrainStep
    ENT, B, T, D = 2, 2, 17, 256
    lcfg = LoraConfig(r=4, lora_alpha=8, lora_dropout=p_drop)
    tok, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = load_models(
        None, ENT, use_lora=True, lora_config=lcfg, unet_config=TINY_CONFIG, vision_config=VIS, text_config=TXT,
        vae_config=dict(block_out_channels=(128, 128, 128, 128), layers_per_block=1), seed=31)
    g = torch.Generator().manual_seed(32)
    for m in unet.modules():
        if isinstance(m, LoRALinear):
            m.lora_B["default"].weight.data.normal_(0, 0.05, generator=g)     # B = 0 at init would zero dA
    from oracle.lora_ref import LoraLinearRef, inject_adapter_in_model_ref
    r_unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_visual_cross_attention_adapter_ref(r_unet, (ENT + 1,))
    # the oracle's LoRA is the INDEPENDENT peft restatement (oracle/lora_ref.py: un-merged base(x) + (alpha / r) B(A(dropout(x)))), not the product's
    inject_adapter_in_model_ref(r_unet, r=lcfg.r, lora_alpha=lcfg.lora_alpha, target_modules=lcfg.target_modules, lora_dropout=0.0)
    r_unet.load_state_dict(unet.state_dict())
    r_txt = CLIPTextModelRef(**TXT).eval(); r_txt.load_state_dict(text_encoder.state_dict())
    r_ia = PhotoVerseAdapterRef(D, 768, ENT + 1).eval(); r_ia.load_state_dict(image_adapter.state_dict())
    r_ta = PhotoVerseAdapterRef(D, 768, ENT + 1).eval(); r_ta.load_state_dict(text_adapter.state_dict())
    for m in (unet, text_encoder, image_adapter, text_adapter):
        m.to("cuda")
    for p in list(r_unet.parameters()) + list(r_txt.parameters()):
        p.requires_grad_(False)
    train_names = [n for n, _ in r_unet.named_parameters() if "to_k_ip" in n or "to_v_ip" in n or "lora_" in n]
    r_params = dict(r_unet.named_parameters())
    for n in train_names:
        r_params[n].requires_grad_(True)

    noisy = torch.randn(B, 4, hw, hw, generator=g)
    noise = torch.randn(B, 4, hw, hw, generator=g)
    timesteps = torch.tensor([731, 42])
    ids = torch.randint(0, 1000, (B, 77), generator=g)
    pidx = torch.tensor([[5], [3]])
    embs = [(torch.randn(B, T, D, generator=g)).half() for _ in range(ENT + 1)]
    forced = [0.1, 0.5, 0.9, 0.4]

    ts = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=hw, w=hw, n_tokens=ENT + 1, clip_tokens=T, clip_dim=D,
                   grad_scale=1024.0, fusion_seed=3)
    out = ts.step(noisy_latents=noisy.cuda(), noise=noise.cuda(), timesteps=timesteps, text_input_ids=ids.cuda(), placeholder_idx=pidx.cuda(),
                  image_embeddings=[e.cuda() for e in embs], forced_fusion=forced)
    torch.cuda.synchronize()

    # ---- oracle: same composition under autograd ----
    masks = {}
    mods = dict(r_unet.named_modules())
    if p_drop > 0:
        from photoverse_amd.ops import Recorder
        assert len(ts.dropout_sites) == 8                  # 4 cross-attention layers x (q, k|v)
        rec = Recorder("cuda")
        got = []
        for site, copies, cols, p in ts.dropout_sites:
            rows = B * 77 if copies == 2 else {320: B * hw * hw, 640: B * hw * hw // 4}[cols]
            got.append((site, copies, cols, rec.dropout(torch.ones(rows, cols, dtype=torch.float16, device="cuda"), p=p, rng=ts.fusion_rng, site=site,
                                                        copies=copies)))
        rec.run()
        torch.cuda.synchronize()
        for site, copies, cols, m in got:
            base = ts.fusion_names[site // 4] + ".transformer_blocks.0.attn2."
            m = m.float().cpu()
            keep = (m > 0).float().mean().item()
            assert abs(keep - (1 - p_drop)) < 0.02 and torch.all((m == 0) | ((m - 1 / (1 - p_drop)).abs() < 1e-3))
            if copies == 1:
                masks[mods[base + "to_q"]] = m
            else:
                masks[mods[base + "to_k"]], masks[mods[base + "to_v"]] = m[:, :cols], m[:, cols:]
        assert not torch.equal(masks[mods[base + "to_k"]], masks[mods[base + "to_v"]])

    # the device draws the dropout masks (counter-based generator); the oracle's LoRA layers apply exactly those masks through their hook
    for mod, msk in masks.items():
        assert isinstance(mod, LoraLinearRef)
        mod.dropout_hook = lambda self, x, msk=msk: (x.reshape(-1, x.shape[-1]) * msk).view_as(x)
    try:
        e32 = [e.float() for e in embs]
        concept = r_ta(e32)
        ehs = r_txt({"text_input_ids": ids, "concept_text_embeddings": concept, "concept_placeholder_idx": pidx})[0]
        ehs_img = r_ia(e32)
        for name, u in zip(ts.fusion_names, forced):
            mods[name + ".transformer_blocks.0.attn2"].processor.forced_fusion_seed = u
        with torch.enable_grad():
            pred = r_unet(noisy, timesteps, encoder_hidden_states=(ehs, ehs_img)).sample
            vn = get_visual_cross_attention_values_norm_ref(r_unet)
            d_loss, c_loss, v_loss = F.mse_loss(pred, noise), concept.abs().mean(), vn.mean()
            loss = d_loss + 0.01 * c_loss + 0.001 * v_loss
            loss.backward()
    finally:
        for mod in masks:
            mod.dropout_hook = None
    assert out["loss"].item() == pytest.approx(loss.item(), rel=5e-3)
    assert out["diffusion_loss"].item() == pytest.approx(d_loss.item(), rel=5e-3)
    assert rel_l2(out["noise_pred"], pred.detach()) < 1e-2

    S = ts.grad_scale
    h_params = dict(unet.named_parameters())

    def group_err(pairs):
        a = torch.cat([(hp.grad.float().cpu() / S).flatten() for hp, _ in pairs])
        # a branch dropped by the fusion draw leaves None in the oracle (zero here)
        b = torch.cat([(rp.grad if rp.grad is not None else torch.zeros_like(rp)).flatten() for _, rp in pairs])
        return rel_l2(a, b)
    ip_pairs = [(h_params[n], r_params[n]) for n in train_names if "_ip" in n]
    la_pairs = [(h_params[n], r_params[n]) for n in train_names if "lora_A" in n]
    lb_pairs = [(h_params[n], r_params[n]) for n in train_names if "lora_B" in n]
    ia_pairs = list(zip(image_adapter.parameters(), r_ia.parameters()))
    ta_pairs = list(zip(text_adapter.parameters(), r_ta.parameters()))
    assert len(ip_pairs) == 8 and len(la_pairs) == 12 and len(lb_pairs) == 12
    errs = {k: group_err(v) for k, v in dict(ip=ip_pairs, lora_A=la_pairs, lora_B=lb_pairs, image_adapter=ia_pairs, text_adapter=ta_pairs).items()}
    print("training-step gradient rel-L2 per group:", errs)
    # fp16 activations + fp16 gradient operands across ~60 layers; the adapters add LeakyReLU / LayerNorm kinks (see the adapter test)
    # (measured: ip 1.9e-3, LoRA 2.8e-3, image / text adapter 1.2e-2 / 5.7e-3 - few rows, so single kink flips weigh more than at full size)
    assert errs["ip"] < 1e-2 and errs["lora_A"] < 1e-2 and errs["lora_B"] < 1e-2, errs
    assert errs["image_adapter"] < 4e-2 and errs["text_adapter"] < 4e-2, errs

    # ---- optimizer: clip per module (train.py:538-541), AdamW (:372-377, :545) ----
    groups = ts.trainable_parameters()
    opt = AdamW([p for grp in groups.values() for p in grp], lr=1e-4, weight_decay=1e-2)
    before = {n: h_params[n].detach().clone() for n in train_names}
    sign_h = torch.cat([torch.sign(h_params[n].grad.float().cpu()).flatten() for n in train_names])
    sign_r = torch.cat([torch.sign(r_params[n].grad if r_params[n].grad is not None else torch.zeros_like(r_params[n])).flatten() for n in train_names])
    norms = opt.step(clip_groups=list(groups.values()), max_norm=1.0, grad_scale=S)
    r_all = [r_params[n] for n in train_names]
    want_norms = [float(torch.nn.utils.clip_grad_norm_(list(r_ta.parameters()), 1.0)), float(torch.nn.utils.clip_grad_norm_(list(r_ia.parameters()), 1.0)),
                  float(torch.nn.utils.clip_grad_norm_(r_all, 1.0))]
    r_opt = torch.optim.AdamW(list(r_ta.parameters()) + list(r_ia.parameters()) + r_all, lr=1e-4, weight_decay=1e-2)
    r_opt.step()
    for got, want in zip(norms, want_norms):
        assert float(got) == pytest.approx(want, rel=5e-2)
    # first AdamW step moves every weight by ~lr * sign(g): compare the updates
    upd_h = torch.cat([(h_params[n].detach().cpu() - before[n].cpu()).flatten() for n in train_names])
    upd_r = torch.cat([(r_params[n].detach() - before[n].cpu()).flatten() for n in train_names])
    upd_err = rel_l2(upd_h, upd_r)
    print(f"first AdamW update vs torch.optim.AdamW on the oracle gradients: rel-L2 {upd_err:.3e}")
    # The first AdamW step moves a weight by ~lr * sign(g) whatever |g| is: an entry whose (near-zero) gradient has the other sign in the fp16 device
    # path flips a FULL +-lr step, and those few entries are the whole unmasked error.  So: the flipped entries are few, and everywhere else the
    # update agrees to the optimizer's own arithmetic (a 2x error in lr, beta, eps, weight decay or the clip scale shows up at O(1) here).
    flipped = sign_h != sign_r
    keep = ~flipped
    masked_err = rel_l2(upd_h[keep], upd_r[keep])
    print(f"sign-flipped gradient entries: {flipped.float().mean().item():.3%}; update rel-L2 over the others: {masked_err:.3e}")
    assert upd_err < 0.15 and flipped.float().mean().item() < 0.02 and masked_err < 2e-2


@pytest.mark.parametrize("smooth_face", [False, True])
def test_training_step_with_face_loss_matches_oracle_autograd(need_gpu, smooth_face):
    """The COMPLETE training iteration of train.py:466-545 including the identity-loss branch (:521-535): for one sample of the batch,
    run_inference(from_noised_image=True, training_mode=True, timesteps=3, guidance_scale=2, token_index=0) - two denoising steps without
    gradient, the last one with gradient (per-layer fusion forced), VAE decode, clamp, ArcFace cosine loss - added to the loss with weight
    0.01; gradients of every trainable group against torch autograd over the fp32 oracle composition of the same steps.
    ``smooth_face``: every PReLU slope of the ArcFace trunk set to 1 - removes the fp16-forward kink noise of that trunk (see
    tests/test_loss_gpu.py) so that the plumbing of the whole chain is checked at the 2 % level of the branch's own contribution."""
    import torch.nn.functional as F
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from oracle.arcface_ref import ArcFaceResNet18Ref, FaceLossRef
    from oracle.clip_ref import CLIPTextModelRef
    from oracle.scheduler_ref import DPMSolverMultistepRef
    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef, get_visual_cross_attention_values_norm_ref, set_visual_cross_attention_adapter_ref
    from oracle.vae_ref import AutoencoderKLDecoderRef
    from photoverse_amd.lora import LoraConfig, LoRALinear, inject_adapter_in_model
    from photoverse_amd.loss import ArcFaceResNet18, FaceLoss
    from photoverse_amd.modeling_utils import load_models
    from photoverse_amd.train import TrainStep
    ENT, B, T, D, NS, STEPS, G = 2, 2, 17, 256, 1, 3, 2.0
    FW = 2.0            # face-loss weight: the reference's 0.01 would bury the branch's gradient below the test's resolution
    VAE = dict(block_out_channels=(128, 128, 256, 256), layers_per_block=1)
    lcfg = LoraConfig(r=4, lora_alpha=8)
    tok, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = load_models(
        None, ENT, use_lora=True, lora_config=lcfg, unet_config=TINY_CONFIG, vision_config=VIS, text_config=TXT, vae_config=VAE, seed=51)
    g = torch.Generator().manual_seed(52)
    for m in unet.modules():
        if isinstance(m, LoRALinear):
            m.lora_B["default"].weight.data.normal_(0, 0.05, generator=g)
    from oracle.lora_ref import inject_adapter_in_model_ref
    r_unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_visual_cross_attention_adapter_ref(r_unet, (ENT + 1,))
    inject_adapter_in_model_ref(r_unet, r=lcfg.r, lora_alpha=lcfg.lora_alpha, target_modules=lcfg.target_modules)   # independent peft restatement
    r_unet.load_state_dict(unet.state_dict())
    r_txt = CLIPTextModelRef(**TXT).eval(); r_txt.load_state_dict(text_encoder.state_dict())
    r_ia = PhotoVerseAdapterRef(D, 768, ENT + 1).eval(); r_ia.load_state_dict(image_adapter.state_dict())
    r_ta = PhotoVerseAdapterRef(D, 768, ENT + 1).eval(); r_ta.load_state_dict(text_adapter.state_dict())
    r_vae = AutoencoderKLDecoderRef(with_encoder=True, **VAE).eval(); r_vae.load_state_dict(vae.state_dict())
    torch.manual_seed(53)
    r_face_net = ArcFaceResNet18Ref().eval()
    for m in r_face_net.modules():                        # calibrated BatchNorm statistics (see tests/test_loss_gpu.py)
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.momentum = 1.0
    if smooth_face:
        for m in r_face_net.modules():
            if isinstance(m, torch.nn.PReLU):
                m.weight.data.fill_(1.0)
    r_face_net.train()
    with torch.no_grad():
        r_face_net(torch.randn(8, 1, 128, 128, generator=g) * 0.5)
    r_face_net.eval()
    face_net = ArcFaceResNet18(); face_net.load_state_dict(r_face_net.state_dict())
    face = FaceLoss("cuda", "arcface", model=face_net)
    r_face = FaceLossRef(r_face_net)
    for m in (unet, text_encoder, image_adapter, text_adapter, vae):
        m.to("cuda")
    for p in list(r_unet.parameters()) + list(r_txt.parameters()) + list(r_vae.parameters()) + list(r_face_net.parameters()):
        p.requires_grad_(False)
    r_params = dict(r_unet.named_parameters())
    train_names = [n for n in r_params if "to_k_ip" in n or "to_v_ip" in n or "lora_" in n]
    for n in train_names:
        r_params[n].requires_grad_(True)

    noisy, noise = torch.randn(B, 4, 16, 16, generator=g), torch.randn(B, 4, 16, 16, generator=g)
    timesteps = torch.tensor([731, 42])
    ids = torch.randint(0, 1000, (B, 77), generator=g)
    pidx = torch.tensor([[5], [3]])
    embs = [torch.randn(B, T, D, generator=g).half() for _ in range(ENT + 1)]
    forced = [0.1, 0.5, 0.9, 0.4]
    # the face-loss branch for sample 1 of the batch
    real = torch.rand(NS, 3, 128, 128, generator=g) * 2 - 1
    start = torch.randn(NS, 4, 16, 16, generator=g)
    emb_c, emb_u = embs[0][1:2], torch.randn(NS, T, D, generator=g).half()
    ids_p, pidx_p = torch.randint(0, 1000, (NS, 77), generator=g), torch.tensor([[4]])
    ids_u = torch.randint(0, 1000, (NS, 77), generator=g)
    forced_u, forced_c = [0.5, 0.5, 0.2, 0.5], [0.5, 0.95, 0.5, 0.5]

    ts = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=16, w=16, n_tokens=ENT + 1, clip_tokens=T, clip_dim=D,
                   grad_scale=1024.0, fusion_seed=3, face_loss=face, vae=vae, noise_scheduler=scheduler, face_samples=NS, face_weight=FW,
                   guidance_scale=G, infer_steps=STEPS, image_size=128)
    fi = dict(pixel_values=real.cuda(), start_latents=start.cuda(), image_embeddings=emb_c.cuda(), uncond_image_embeddings=emb_u.cuda(),
              text_input_ids=ids_p.cuda(), placeholder_idx=pidx_p.cuda(), uncond_input_ids=ids_u.cuda(), forced_fusion=(forced_u, forced_c))
    out = ts.step(noisy_latents=noisy.cuda(), noise=noise.cuda(), timesteps=timesteps, text_input_ids=ids.cuda(), placeholder_idx=pidx.cuda(),
                  image_embeddings=[e.cuda() for e in embs], forced_fusion=forced, face_inputs=fi)
    torch.cuda.synchronize()

    mods = dict(r_unet.named_modules())

    def force(vals):
        for name, u in zip(ts.fusion_names, vals):
            mods[name + ".transformer_blocks.0.attn2"].processor.forced_fusion_seed = u
    try:
        e32 = [e.float() for e in embs]
        concept = r_ta(e32)
        ehs = r_txt({"text_input_ids": ids, "concept_text_embeddings": concept, "concept_placeholder_idx": pidx})[0]
        ehs_img = r_ia(e32)
        force(forced)
        with torch.enable_grad():
            pred = r_unet(noisy, timesteps, encoder_hidden_states=(ehs, ehs_img)).sample
            vn = get_visual_cross_attention_values_norm_ref(r_unet)
            main_loss = F.mse_loss(pred, noise) + 0.01 * concept.abs().mean() + 0.001 * vn.mean()
            # ---- run_inference(..., token_index=0, from_noised_image=True, training_mode=True) (infer.py:70-123)
            c0 = r_ta([emb_c.float()], token_index=0)
            text_c = r_txt({"text_input_ids": ids_p, "concept_text_embeddings": c0, "concept_placeholder_idx": pidx_p})[0]
            ip_c, ip_u = r_ia([emb_c.float()], token_index=0), r_ia([emb_u.float()], token_index=0)
            text_u = r_txt({"text_input_ids": ids_u})[0]
            sch = DPMSolverMultistepRef()
            sch.set_timesteps(STEPS)
            lat = start * sch.init_noise_sigma
            for i, t in enumerate(sch.timesteps):
                last = i == len(sch.timesteps) - 1
                with torch.set_grad_enabled(last):
                    if last:
                        force(forced_u)
                    eps_u = r_unet(lat, t, encoder_hidden_states=(text_u, ip_u)).sample
                    if last:
                        force(forced_c)
                    eps_c = r_unet(lat, t, encoder_hidden_states=(text_c, ip_c)).sample
                    lat = sch.step(eps_u + G * (eps_c - eps_u), t, lat)
            images = r_vae.decode(lat / 0.18215).sample.clamp(-1, 1)
            floss = r_face(real, images, normalize=False)
            loss = main_loss + FW * floss
            plist = [r_params[n] for n in train_names] + list(r_ia.parameters()) + list(r_ta.parameters())
            main_only = torch.autograd.grad(main_loss, plist, retain_graph=True, allow_unused=True)
            loss.backward()
    finally:
        pass
    print(f"face branch: floss {out['face_loss'].item():.5f} vs {floss.item():.5f}; images rel-L2 {rel_l2(out['face_images'], images.detach()):.3e}")
    assert rel_l2(out["face_images"], images.detach()) < 2e-2
    assert out["face_loss"].item() == pytest.approx(floss.item(), rel=3e-2, abs=2e-3)
    assert out["loss"].item() == pytest.approx(loss.item(), rel=5e-3)
    S = ts.grad_scale
    h_params = dict(unet.named_parameters())

    def group_err(pairs):
        a = torch.cat([(hp.grad.float().cpu() / S).flatten() for hp, _ in pairs])
        b = torch.cat([(rp.grad if rp.grad is not None else torch.zeros_like(rp)).flatten() for _, rp in pairs])
        return rel_l2(a, b)
    groups = dict(ip=[(h_params[n], r_params[n]) for n in train_names if "_ip" in n],
                  lora_A=[(h_params[n], r_params[n]) for n in train_names if "lora_A" in n],
                  lora_B=[(h_params[n], r_params[n]) for n in train_names if "lora_B" in n],
                  image_adapter=list(zip(image_adapter.parameters(), r_ia.parameters())),
                  text_adapter=list(zip(text_adapter.parameters(), r_ta.parameters())))
    errs = {k: group_err(v) for k, v in groups.items()}
    print("training step + face loss, gradient rel-L2 per group:", errs)
    # with the branch dominating the gradient (FW = 2) every group inherits part of the ArcFace trunk's PReLU / max-pool kink noise
    # (6-8e-2 on the image gradient in isolation, tests/test_loss_gpu.py); with FW = 0.01 the same run gives 1.4e-3 ... 1.2e-2
    # measured: 2.2-2.6e-2 (real trunk, branch = 23 % of the gradient); 2.3e-3 ... 1.2e-2 (smooth trunk, branch = 8 %)
    # round 4 box: 3.3e-2 max (real trunk), 1.24e-2 max (smooth trunk) - bounds at ~1.5x
    assert max(errs.values()) < (1.9e-2 if smooth_face else 5e-2), errs
    # the branch's own contribution must be far above those errors, or the comparison would not see it
    tot = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in plist])
    mo = torch.cat([(gm if gm is not None else torch.zeros_like(p)).flatten() for gm, p in zip(main_only, plist)])
    contrib = rel_l2(mo, tot)
    print(f"gradient change from the face-loss branch: {contrib:.3f} (rel-L2 of the total gradient)")
    assert contrib > (0.05 if smooth_face else 0.2)


@pytest.mark.parametrize("with_face", [False, True])
def test_training_iteration_loop_body(need_gpu, with_face):
    """train.py:464-549 end to end on the package's own modules (VAE encode, CLIP image encoder, tokenizer, TrainStep, AdamW): three
    iterations on one batch - finite losses, every trainable parameter group moves, the loss goes down on the repeated batch."""
    from types import SimpleNamespace
    from photoverse_amd.lora import LoraConfig
    from photoverse_amd.loss import FaceLoss
    from photoverse_amd.modeling_utils import load_models
    from photoverse_amd.optim import AdamW
    from photoverse_amd.train import TrainStep, training_iteration
    from oracle.unet_ref import TINY_CONFIG
    ENT, B = 2, 2
    VAE = dict(block_out_channels=(128, 128, 256, 256), layers_per_block=1)
    tok, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = load_models(
        None, ENT, use_lora=True, lora_config=LoraConfig(r=4, lora_alpha=4, lora_dropout=0.1), unet_config=TINY_CONFIG, vision_config=VIS,
        text_config=TXT, vae_config=VAE, seed=61)
    for m in (unet, text_encoder, image_adapter, text_adapter, vae, image_encoder):
        m.to("cuda")
    face = FaceLoss("cuda", "arcface") if with_face else None
    step = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=16, w=16, n_tokens=ENT + 1, clip_tokens=17, clip_dim=256,
                     grad_scale=1024.0, fusion_seed=5, face_loss=face, vae=vae if with_face else None, noise_scheduler=scheduler, face_samples=1,
                     infer_steps=3, image_size=128)
    groups = step.trainable_parameters()
    opt = AdamW([p for g_ in groups.values() for p in g_], lr=2e-4, weight_decay=1e-2)
    g = torch.Generator().manual_seed(62)
    batch = {"pixel_values": torch.rand(B, 3, 128, 128, generator=g) * 2 - 1, "pixel_values_clip": torch.randn(B, 3, 56, 56, generator=g),
             "text_input_ids": torch.randint(0, 1000, (B, 77), generator=g), "concept_placeholder_idx": torch.tensor([[5], [3]])}
    before = {k: [p.detach().clone() for p in v] for k, v in groups.items()}
    losses = []
    for it in range(3):
        out = training_iteration(step, opt, batch, tok, image_encoder, vae, scheduler, "cuda", [1, 2], ENT, generator=torch.Generator().manual_seed(63))
        losses.append(float(out["loss"]))
        assert all(torch.isfinite(n).all() for n in out["grad_norms"])
        if with_face:
            assert torch.isfinite(out["face_loss"]).all() and out["face_images"].shape == (1, 3, 128, 128)
    assert all(l == l and abs(l) < 1e4 for l in losses)
    for k, ps in groups.items():
        moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(ps, before[k]))
        assert moved > 0, k
    assert losses[-1] < losses[0]              # same batch, same draws (generator re-seeded): three AdamW steps lower its loss


def test_adamw_state_dict_is_interchangeable_with_torch(need_gpu):
    """save_progress stores optimizer.state_dict() (modeling_utils.py:43-44): the HIP optimizer writes / reads torch.optim.AdamW's format -
    one step here, state moved into torch.optim.AdamW (and back), a second step on both sides gives the same parameters."""
    from photoverse_amd.optim import AdamW
    g = torch.Generator().manual_seed(9)
    ps = [torch.nn.Parameter(torch.randn(s_, generator=g).cuda()) for s_ in ((64, 32), (32,), (5, 7))]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    o1 = AdamW(ps, lr=1e-3, weight_decay=1e-2)
    for p in ps:
        p.grad = torch.randn(p.shape, generator=g).cuda()
    o1.step()
    for p, q in zip(ps, qs):
        q.data.copy_(p.data)
    o2 = torch.optim.AdamW(qs, lr=5e-2)
    import copy
    o2.load_state_dict(copy.deepcopy(o1.state_dict()))    # (a file round trip; torch keeps same-device tensors by reference) - also carries lr / betas / ...
    assert o2.param_groups[0]["lr"] == 1e-3
    for p, q in zip(ps, qs):
        gr = torch.randn(p.shape, generator=g).cuda()
        p.grad, q.grad = gr.clone(), gr.clone()
    o1.step()
    o2.step()
    for p, q in zip(ps, qs):
        assert rel_l2(p.detach(), q.detach()) < 1e-6
    o3 = AdamW([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1.0)
    o3.load_state_dict(copy.deepcopy(o2.state_dict()))
    assert o3.step_count == 2 and o3.lr == 1e-3
    assert torch.equal(o3.state[id(o3.params[0])][0], o2.state[qs[0]]["exp_avg"])


def test_full_size_identity_loss_branch_matches_oracle_autograd(need_gpu, full_weights):
    """The identity-loss branch at the FULL model sizes: SD-v1.5 UNet, 12-layer text encoder, 1024-wide adapters, the real VAE decoder
    (32x32 latents -> 256x256 image, mid-block attention over 1024 tokens of width 512), ArcFace IR-ResNet18 at 128x128; B = 1, one face
    sample, two inference steps (one without, one with gradient), guidance 2.  Image, loss and every gradient group against torch autograd
    over the LIVE fp32 oracle composition (the device draws the posterior / fusion samples, so this one cannot be a fixture): a quarter of the
    512x512 spatial size keeps the oracle's autograd at ~30 s of host time - the 64x64-latent shapes (N = 4096 attention backward, every level)
    are covered by tests/test_fullsize_gpu.py::test_full_size_training_gradients and the 512x512 decode by test_whole_generation_full_size."""
    import torch.nn.functional as F
    from types import SimpleNamespace
    from oracle.adapters_ref import PhotoVerseAdapterRef
    from oracle.arcface_ref import ArcFaceResNet18Ref, FaceLossRef
    from oracle.clip_ref import CLIPTextModelRef
    from oracle.scheduler_ref import DPMSolverMultistepRef
    from oracle.unet_ref import UNet2DConditionModelRef, get_visual_cross_attention_values_norm_ref, set_visual_cross_attention_adapter_ref
    from oracle.vae_ref import AutoencoderKLDecoderRef
    from photoverse_amd.adapters import PhotoVerseAdapter
    from photoverse_amd.clip import CLIPTextModel
    from photoverse_amd.lora import LoraConfig, LoRALinear, inject_adapter_in_model
    from photoverse_amd.loss import ArcFaceResNet18, FaceLoss
    from photoverse_amd.scheduler import DPMSolverMultistepScheduler
    from photoverse_amd.train import TrainStep
    from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter
    from photoverse_amd.vae import AutoencoderKL
    from oracle import fullsize as fs
    torch.manual_seed(0)
    E, B, T, D, STEPS, G, FW = 5, 1, 257, 1024, 2, 2.0, 2.0
    LAT, IMG = 32, 256
    lcfg = LoraConfig(r=8, lora_alpha=1)
    from oracle.lora_ref import LoraLinearRef, inject_adapter_in_model_ref
    with fs.no_init():                                      # the session's seeded full-size weights (tests/conftest.py): no 40 s of default inits
        r_unet = UNet2DConditionModelRef().eval()
        set_visual_cross_attention_adapter_ref(r_unet, (E,))
    r_unet.load_state_dict(full_weights("unet"))
    inject_adapter_in_model_ref(r_unet, r=lcfg.r, lora_alpha=lcfg.lora_alpha, target_modules=lcfg.target_modules)   # independent peft restatement
    g = torch.Generator().manual_seed(71)
    for m in r_unet.modules():
        if isinstance(m, LoraLinearRef):
            m.lora_B["default"].weight.data.normal_(0, 0.05, generator=g)
    with fs.no_init():
        r_txt = CLIPTextModelRef().eval()
        r_ia, r_ta = PhotoVerseAdapterRef(D, 768, E).eval(), PhotoVerseAdapterRef(D, 768, E).eval()
        r_vae = AutoencoderKLDecoderRef().eval()
    r_txt.load_state_dict(full_weights("text")); r_ia.load_state_dict(full_weights("image_adapter")); r_ta.load_state_dict(full_weights("text_adapter"))
    r_vae.load_state_dict({k: v for k, v in full_weights("vae").items() if k.startswith(("decoder.", "post_quant_conv."))})
    r_face_net = ArcFaceResNet18Ref().eval()
    for m in r_face_net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
            m.momentum = 1.0
    r_face_net.train()
    with torch.no_grad():
        r_face_net(torch.randn(8, 1, 128, 128, generator=g) * 0.5)
    r_face_net.eval()
    with fs.no_init():
        unet = UNet2DConditionModel()
        set_visual_cross_attention_adapter(unet, (E,))
        inject_adapter_in_model(lcfg, unet)
        text_encoder = CLIPTextModel()
        image_adapter, text_adapter = PhotoVerseAdapter(D, 768, E), PhotoVerseAdapter(D, 768, E)
        vae = AutoencoderKL(with_encoder=False) if "with_encoder" in AutoencoderKL.__init__.__code__.co_varnames else AutoencoderKL()
    unet.load_state_dict(r_unet.state_dict())
    text_encoder.load_state_dict(r_txt.state_dict())
    image_adapter.load_state_dict(r_ia.state_dict())
    text_adapter.load_state_dict(r_ta.state_dict())
    vae.load_state_dict(r_vae.state_dict(), strict=False)
    face_net = ArcFaceResNet18(); face_net.load_state_dict(r_face_net.state_dict())
    for m in (unet, text_encoder, image_adapter, text_adapter, vae):
        m.to("cuda")
    face, r_face = FaceLoss("cuda", "arcface", model=face_net), FaceLossRef(r_face_net)
    for p in list(r_unet.parameters()) + list(r_txt.parameters()) + list(r_vae.parameters()) + list(r_face_net.parameters()):
        p.requires_grad_(False)
    r_params = dict(r_unet.named_parameters())
    train_names = [n for n in r_params if "to_k_ip" in n or "to_v_ip" in n or "lora_" in n]
    for n in train_names:
        r_params[n].requires_grad_(True)
    noisy, noise = torch.randn(B, 4, LAT, LAT, generator=g), torch.randn(B, 4, LAT, LAT, generator=g)
    timesteps = torch.tensor([417])
    ids, pidx = torch.randint(0, 49000, (B, 77), generator=g), torch.tensor([[4]])
    embs = [torch.randn(B, T, D, generator=g).half() for _ in range(E)]
    forced = [0.5] * 16
    real = torch.rand(1, 3, IMG, IMG, generator=g) * 2 - 1
    start = torch.randn(1, 4, LAT, LAT, generator=g)
    emb_u = torch.randn(1, T, D, generator=g).half()
    ids_p, ids_u = torch.randint(0, 49000, (1, 77), generator=g), torch.randint(0, 49000, (1, 77), generator=g)
    forced_u, forced_c = [0.5] * 16, [0.5] * 16
    forced_c[2], forced_u[7] = 0.05, 0.95
    ts = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=LAT, w=LAT, n_tokens=E, grad_scale=4096.0, fusion_seed=3, face_loss=face,
                   vae=vae, noise_scheduler=SimpleNamespace(config=DPMSolverMultistepScheduler().config), face_samples=1, face_weight=FW,
                   guidance_scale=G, infer_steps=STEPS, use_graph=False)
    fi = dict(pixel_values=real.cuda(), start_latents=start.cuda(), image_embeddings=embs[0].cuda(), uncond_image_embeddings=emb_u.cuda(),
              text_input_ids=ids_p.cuda(), placeholder_idx=pidx.cuda(), uncond_input_ids=ids_u.cuda(), forced_fusion=(forced_u, forced_c))
    out = ts.step(noisy_latents=noisy.cuda(), noise=noise.cuda(), timesteps=timesteps, text_input_ids=ids.cuda(), placeholder_idx=pidx.cuda(),
                  image_embeddings=[e.cuda() for e in embs], forced_fusion=forced, face_inputs=fi)
    torch.cuda.synchronize()
    mods = dict(r_unet.named_modules())

    def force(vals):
        for name, u in zip(ts.fusion_names, vals):
            mods[name + ".transformer_blocks.0.attn2"].processor.forced_fusion_seed = u
    try:
        e32 = [e.float() for e in embs]
        concept = r_ta(e32)
        ehs = r_txt({"text_input_ids": ids, "concept_text_embeddings": concept, "concept_placeholder_idx": pidx})[0]
        ehs_img = r_ia(e32)
        force(forced)
        with torch.enable_grad():
            pred = r_unet(noisy, timesteps, encoder_hidden_states=(ehs, ehs_img)).sample
            vn = get_visual_cross_attention_values_norm_ref(r_unet)
            main_loss = F.mse_loss(pred, noise) + 0.01 * concept.abs().mean() + 0.001 * vn.mean()
            c0 = r_ta([e32[0]], token_index=0)
            text_c = r_txt({"text_input_ids": ids_p, "concept_text_embeddings": c0, "concept_placeholder_idx": pidx})[0]
            ip_c, ip_u = r_ia([e32[0]], token_index=0), r_ia([emb_u.float()], token_index=0)
            text_u = r_txt({"text_input_ids": ids_u})[0]
            sch = DPMSolverMultistepRef()
            sch.set_timesteps(STEPS)
            lat = start * sch.init_noise_sigma
            for i, t in enumerate(sch.timesteps):
                last = i == len(sch.timesteps) - 1
                with torch.set_grad_enabled(last):
                    if last:
                        force(forced_u)
                    eps_u = r_unet(lat, t, encoder_hidden_states=(text_u, ip_u)).sample
                    if last:
                        force(forced_c)
                    eps_c = r_unet(lat, t, encoder_hidden_states=(text_c, ip_c)).sample
                    lat = sch.step(eps_u + G * (eps_c - eps_u), t, lat)
            images = r_vae.decode(lat / 0.18215).sample.clamp(-1, 1)
            floss = r_face(real, images, normalize=False)
            loss = main_loss + FW * floss
            loss.backward()
    finally:
        pass
    err_img = rel_l2(out["face_images"], images.detach())
    print(f"full-size face branch: floss {out['face_loss'].item():.5f} vs {floss.item():.5f}; {IMG}x{IMG} image rel-L2 {err_img:.3e}")
    assert err_img < 2e-2
    assert out["face_loss"].item() == pytest.approx(floss.item(), rel=3e-2, abs=3e-3)
    S = ts.grad_scale
    h_params = dict(unet.named_parameters())

    def group_err(pairs):
        a = torch.cat([(hp.grad.float().cpu() / S).flatten() for hp, _ in pairs])
        b = torch.cat([(rp.grad if rp.grad is not None else torch.zeros_like(rp)).flatten() for _, rp in pairs])
        return rel_l2(a, b)
    groups = dict(ip=[(h_params[n], r_params[n]) for n in train_names if "_ip" in n],
                  lora_A=[(h_params[n], r_params[n]) for n in train_names if "lora_A" in n],
                  lora_B=[(h_params[n], r_params[n]) for n in train_names if "lora_B" in n],
                  image_adapter=list(zip(image_adapter.parameters(), r_ia.parameters())),
                  text_adapter=list(zip(text_adapter.parameters(), r_ta.parameters())))
    errs = {k: group_err(v) for k, v in groups.items()}
    print("full-size training step + face loss, gradient rel-L2 per group:", errs)
    # measured (round 4): ip 2.4e-2, LoRA 1.0-1.4e-2, image adapter 2.9e-2, text adapter 1.3e-2 - the ArcFace trunk's PReLU / max-pool kinks
    # under an fp16 forward (6-8e-2 on the image gradient in isolation) dominate every group; bound at ~1.6x the largest
    assert max(errs.values()) < 4.8e-2, errs


@pytest.mark.parametrize("face", [False, True])
def test_train_cli_runs_and_writes_reference_layout_checkpoints(need_gpu, tmp_path, face):
    """``train.py`` (the counterpart of the reference's training CLI) as a subprocess on the tiny config with synthetic data: three steps
    with LoRA (reference defaults incl. dropout) and, for ``face``, the ArcFace identity loss; the step and final checkpoints load through
    ``load_photoverse_model`` (reference layout, modeling_utils.py:13-50) and carry the optimizer state."""
    import subprocess
    import sys
    from photoverse_amd.modeling_utils import load_models, load_photoverse_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "train.py"), "--pretrained_model_name_or_path", "random", "--tiny", "--synthetic_data",
           "--max_train_steps", "3", "--train_batch_size", "2", "--resolution", "128", "--extra_num_tokens", "2", "--image_encoder_layers_idx", "1", "2",
           "--use_lora", "--checkpoint_save_steps", "2", "--samples_save_steps", "2", "--denoise_timesteps", "3", "--num_of_samples_to_save", "2",
           "--output_dir", str(tmp_path), "--seed", "7", "--lr_scheduler", "constant_with_warmup",
           "--lr_warmup_steps", "2", "--learning_rate", "1e-4"] + (["--face_loss", "arcface"] if face else ["--gradient_accumulation_steps", "2"])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("step ") and "loss_mle=" in l]
    assert len(lines) == 3 and ("loss_face=" in lines[0]) == face
    assert "lr=0," in lines[0].replace(" ", "") or "lr=0" in lines[0]          # warm-up: first step at lr 0, then 5e-5, then 1e-4
    assert os.path.exists(tmp_path / "photoverse_000002.pt") and os.path.exists(tmp_path / "photoverse.pt")
    from PIL import Image
    grid = Image.open(tmp_path / "00002.jpg")               # input | condition | generated, two samples, 50-pixel title strip
    assert grid.size == (3 * 128, 2 * 128 + 50)
    assert ("face_similarity=" in r.stdout) == face
    sd = torch.load(tmp_path / "photoverse.pt", map_location="cpu")
    assert set(sd) >= {"image_adapter", "text_adapter", "cross_attention_adapter", "lora_config", "optimizer"}
    assert any("lora_A" in k for k in sd["cross_attention_adapter"]) and any("to_k_ip" in k for k in sd["cross_attention_adapter"])
    assert len(sd["optimizer"]["state"]) > 0 and int(sd["optimizer"]["state"][0]["step"]) == 3
    from oracle.unet_ref import TINY_CONFIG
    _, _, _, unet, _, ia, ta, _, _ = load_models(None, 2, unet_config=TINY_CONFIG, vision_config=VIS, text_config=TXT,
                                                  vae_config=dict(block_out_channels=(128, 128, 128, 128), layers_per_block=1), seed=1)
    ia, ta, unet, lcfg = load_photoverse_model(str(tmp_path / "photoverse.pt"), ia, ta, unet)
    assert lcfg is not None and lcfg.r == 8 and lcfg.lora_dropout == pytest.approx(0.1)


@pytest.mark.parametrize("B,ent,use_lora,face,guidance", [(1, 0, False, False, 2.0), (3, 1, False, True, 1.0), (2, 4, True, True, 7.5)])
def test_training_step_configurations_run(need_gpu, B, ent, use_lora, face, guidance):
    """Shapes / options beyond the parity tests: one token (extra_num_tokens = 0), odd batch, no LoRA (only adapters + to_k_ip / to_v_ip
    train), guidance 1 (the uncond forward of the last step carries no gradient), five tokens with a large guidance: finite losses,
    a gradient on every trainable parameter, nothing on frozen ones."""
    from photoverse_amd.lora import LoraConfig
    from photoverse_amd.loss import FaceLoss
    from photoverse_amd.modeling_utils import load_models
    from photoverse_amd.train import TrainStep
    from oracle.unet_ref import TINY_CONFIG
    VAE = dict(block_out_channels=(128, 128, 256, 256), layers_per_block=1)
    vis = dict(VIS, num_hidden_layers=max(ent, 1) + 1)
    tok, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = load_models(
        None, ent, use_lora=use_lora, lora_config=LoraConfig(r=4, lora_alpha=4, lora_dropout=0.1) if use_lora else None, unet_config=TINY_CONFIG,
        vision_config=vis, text_config=TXT, vae_config=VAE, seed=81)
    for m in (unet, text_encoder, image_adapter, text_adapter, vae):
        m.to("cuda")
    E = ent + 1
    step = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=16, w=16, n_tokens=E, clip_tokens=17, clip_dim=256, grad_scale=512.0,
                     face_loss=FaceLoss("cuda", "arcface") if face else None, vae=vae if face else None, noise_scheduler=scheduler, face_samples=1,
                     guidance_scale=guidance, infer_steps=2, image_size=128)
    g = torch.Generator().manual_seed(82)
    fi = None
    if face:
        fi = dict(pixel_values=(torch.rand(1, 3, 128, 128, generator=g) * 2 - 1).cuda(), start_latents=torch.randn(1, 4, 16, 16, generator=g).cuda(),
                  image_embeddings=torch.randn(1, 17, 256, generator=g).half().cuda(), uncond_image_embeddings=torch.randn(1, 17, 256, generator=g).half().cuda(),
                  text_input_ids=torch.randint(0, 1000, (1, 77), generator=g).cuda(), placeholder_idx=torch.tensor([[4]]).cuda(),
                  uncond_input_ids=torch.randint(0, 1000, (1, 77), generator=g).cuda())
    for it in range(2):                                   # second iteration replays the captured graphs
        out = step.step(noisy_latents=torch.randn(B, 4, 16, 16, generator=g).cuda(), noise=torch.randn(B, 4, 16, 16, generator=g).cuda(),
                        timesteps=torch.randint(0, 1000, (B,), generator=g), text_input_ids=torch.randint(0, 1000, (B, 77), generator=g).cuda(),
                        placeholder_idx=torch.randint(1, 70, (B, 1), generator=g).cuda(),
                        image_embeddings=[torch.randn(B, 17, 256, generator=g).half().cuda() for _ in range(E)], face_inputs=fi)
        torch.cuda.synchronize()
        assert torch.isfinite(out["loss"]).all()
    groups = step.trainable_parameters()
    n_lora = sum(1 for n, _ in unet.named_parameters() if "lora_" in n)
    assert (n_lora > 0) == use_lora and len(groups["unet"]) == 8 + n_lora
    for k, ps in groups.items():
        for p in ps:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
    trainable = {id(p) for ps in groups.values() for p in ps}
    assert all(p.grad is None for p in unet.parameters() if id(p) not in trainable)


def test_training_step_is_bit_reproducible(need_gpu):
    """No atomics and fixed-order reductions everywhere (split-K slabs, GroupNorm / LayerNorm partials, attention backward, weight-gradient
    split-K, the optimizer's norms): the same inputs with the same forced fusion draws give bit-identical losses and gradients - eager
    first iteration vs the HIP-graph replays of the next two - with the identity-loss branch switched on."""
    from photoverse_amd.lora import LoraConfig
    from photoverse_amd.loss import FaceLoss
    from photoverse_amd.modeling_utils import load_models
    from photoverse_amd.train import TrainStep
    from oracle.unet_ref import TINY_CONFIG
    VAE = dict(block_out_channels=(128, 128, 256, 256), layers_per_block=1)
    tok, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = load_models(
        None, 2, use_lora=True, lora_config=LoraConfig(r=4, lora_alpha=4, lora_dropout=0.0), unet_config=TINY_CONFIG, vision_config=VIS,
        text_config=TXT, vae_config=VAE, seed=91)
    for m in (unet, text_encoder, image_adapter, text_adapter, vae):
        m.to("cuda")
    B, E = 2, 3
    step = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=16, w=16, n_tokens=E, clip_tokens=17, clip_dim=256, grad_scale=512.0,
                     face_loss=FaceLoss("cuda", "arcface"), vae=vae, noise_scheduler=scheduler, face_samples=1, infer_steps=3, image_size=128)
    g = torch.Generator().manual_seed(92)
    fi = dict(pixel_values=(torch.rand(1, 3, 128, 128, generator=g) * 2 - 1).cuda(), start_latents=torch.randn(1, 4, 16, 16, generator=g).cuda(),
              image_embeddings=torch.randn(1, 17, 256, generator=g).half().cuda(), uncond_image_embeddings=torch.randn(1, 17, 256, generator=g).half().cuda(),
              text_input_ids=torch.randint(0, 1000, (1, 77), generator=g).cuda(), placeholder_idx=torch.tensor([[4]]).cuda(),
              uncond_input_ids=torch.randint(0, 1000, (1, 77), generator=g).cuda(), forced_fusion=([0.5, 0.1, 0.9, 0.5], [0.9, 0.5, 0.5, 0.1]))
    kw = dict(noisy_latents=torch.randn(B, 4, 16, 16, generator=g).cuda(), noise=torch.randn(B, 4, 16, 16, generator=g).cuda(),
              timesteps=torch.tensor([5, 900]), text_input_ids=torch.randint(0, 1000, (B, 77), generator=g).cuda(),
              placeholder_idx=torch.tensor([[2], [9]]).cuda(), image_embeddings=[torch.randn(B, 17, 256, generator=g).half().cuda() for _ in range(E)],
              forced_fusion=[0.1, 0.5, 0.9, 0.5], face_inputs=fi)
    params = [p for ps in step.trainable_parameters().values() for p in ps]
    runs = []
    for it in range(3):
        out = step.step(**kw)
        torch.cuda.synchronize()
        runs.append((out["loss"].clone(), out["face_loss"].clone(), [p.grad.detach().clone() for p in params]))
    assert step.graph is not None and step.face.graph is not None
    for it in (1, 2):
        assert torch.equal(runs[it][0], runs[0][0]) and torch.equal(runs[it][1], runs[0][1])
        assert all(torch.equal(a, b) for a, b in zip(runs[it][2], runs[0][2])), it


def test_gradient_accumulation_sums_micro_batches(need_gpu):
    """accelerator.accumulate (train.py:464, --gradient_accumulation_steps): two micro-batches through step(accumulate=...) leave the SUM of
    their gradients (the optimizer divides by the count through grad_scale); compared with the two gradients taken separately."""
    from photoverse_amd.lora import LoraConfig
    from photoverse_amd.modeling_utils import load_models
    from photoverse_amd.train import TrainStep, _detach_grads
    from oracle.unet_ref import TINY_CONFIG
    tok, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, _ = load_models(
        None, 1, use_lora=True, lora_config=LoraConfig(r=4, lora_alpha=4, lora_dropout=0.0), unet_config=TINY_CONFIG, vision_config=VIS,
        text_config=TXT, vae_config=dict(block_out_channels=(128, 128, 128, 128), layers_per_block=1), seed=95)
    for m in (unet, text_encoder, image_adapter, text_adapter):
        m.to("cuda")
    B, E = 2, 2
    step = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=16, w=16, n_tokens=E, clip_tokens=17, clip_dim=256, grad_scale=512.0)
    g = torch.Generator().manual_seed(96)
    mk = lambda: dict(noisy_latents=torch.randn(B, 4, 16, 16, generator=g).cuda(), noise=torch.randn(B, 4, 16, 16, generator=g).cuda(),
                      timesteps=torch.randint(0, 1000, (B,), generator=g), text_input_ids=torch.randint(0, 1000, (B, 77), generator=g).cuda(),
                      placeholder_idx=torch.randint(1, 70, (B, 1), generator=g).cuda(),
                      image_embeddings=[torch.randn(B, 17, 256, generator=g).half().cuda() for _ in range(E)], forced_fusion=[0.5, 0.1, 0.9, 0.5])
    a, b = mk(), mk()
    params = [p for ps in step.trainable_parameters().values() for p in ps]
    step.step(**a)
    ga = [p.grad.detach().clone() for p in params]
    step.step(**b)
    gb = [p.grad.detach().clone() for p in params]
    step.step(**a)                                         # first micro-batch of an update ...
    _detach_grads(step)
    step.step(**b, accumulate=True)                        # ... second one adds
    torch.cuda.synchronize()
    for p, x, y in zip(params, ga, gb):
        assert rel_l2(p.grad, x + y) < 1e-6


def test_data_parallel_training_matches_full_batch(need_gpu, tmp_path):
    """Data-parallel training (the reference trains under accelerate / DDP, train.py:299-305, :398-400): two ranks (two processes sharing this
    GPU, gloo rendezvous) each step HALF of a seeded global batch, ``GradientReducer`` sums the gradients, AdamW takes the mean through
    ``grad_scale``.  Both ranks must end with IDENTICAL gradients and parameters, and these must match one process stepping the whole batch
    (the losses are batch means, so the mean of the half-batch gradients is the full-batch gradient)."""
    import socket
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _ddp_train_worker as W
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ddp_train_worker.py")
    B, world = 2, 2
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path), str(B)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    full = W.one_step(W.build(101), W.rows(W.global_batch(B * world, 2, 102), 0, B * world), B * world, lambda params: None)
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out[-3000:]
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    assert r0["world"] == r1["world"] == 2 and full["world"] == 1
    assert all(torch.equal(a, b) for a, b in zip(r0["grads"], r1["grads"]))               # one sum, the same bits on every rank
    assert all(torch.equal(a, b) for a, b in zip(r0["params"], r1["params"]))
    assert not torch.equal(r0["loss"], r1["loss"])                                          # ... from different halves of the batch
    num = sum(float((a - b).pow(2).sum()) for a, b in zip(r0["grads"], full["grads"]))
    den = sum(float(b.pow(2).sum()) for b in full["grads"])
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5
    assert abs(float(0.5 * (r0["loss"] + r1["loss"]) - full["loss"])) < 2e-3 * abs(float(full["loss"]))
    moved = sum(float((a - b).abs().max()) for a, b in zip(r0["params"], full["params"]))
    assert moved < 1e-3 * len(full["params"])                                               # the same AdamW update (lr 1e-3, sign-like first step)


def test_train_cli_two_ranks(need_gpu, tmp_path):
    """The training CLI launched as two ranks (both on this GPU, gloo instead of RCCL): every rank runs its own batches, the gradients are
    all-reduced once per optimizer step, only rank 0 logs and writes the checkpoint, and the run ends cleanly on both."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, os.path.join(root, "train.py"), "--pretrained_model_name_or_path", "random", "--tiny", "--synthetic_data",
           "--max_train_steps", "2", "--train_batch_size", "2", "--resolution", "128", "--extra_num_tokens", "2", "--image_encoder_layers_idx", "1", "2",
           "--use_lora", "--checkpoint_save_steps", "100", "--samples_save_steps", "0", "--output_dir", str(tmp_path), "--seed", "7",
           "--gradient_accumulation_steps", "2"]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PV_DIST_BACKEND="gloo")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, o[-2000:] + e[-2000:]
        outs.append(o)
    assert len([l for l in outs[0].splitlines() if l.startswith("step ") and "loss_mle=" in l]) == 2 and "saved " in outs[0]
    assert "step " not in outs[1] and "saved " not in outs[1]
    sd = torch.load(tmp_path / "photoverse.pt", map_location="cpu")
    assert int(sd["optimizer"]["state"][0]["step"]) == 2


def test_gradient_reducer_over_rccl_one_rank(need_gpu):
    """The flat gradient bucket through RCCL itself (backend "nccl", a one-rank group on this GPU; the two-rank tests above use gloo because two
    RCCL ranks cannot share a device): the all-reduce runs on device memory, the gradients come back as views of the bucket, unchanged."""
    import socket
    import torch.distributed as dist
    from photoverse_amd.train import GradientReducer
    if dist.is_initialized():
        pytest.skip("a process group is already initialised in this process")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g = torch.Generator().manual_seed(11)
        params = [torch.nn.Parameter(torch.zeros(s, device="cuda")) for s in ((640, 768), (8, 320), (1024,))]
        grads = [torch.randn(p.shape, generator=g).cuda() for p in params]
        for p, x in zip(params, grads):
            p.grad = x.clone()
        red = GradientReducer(params, force=True)
        assert red() == 1
        torch.cuda.synchronize()
        for p, x in zip(params, grads):
            assert torch.equal(p.grad, x) and p.grad.data_ptr() >= red.flat.data_ptr()
    finally:
        dist.destroy_process_group()


def test_full_size_training_step_batch_replication(need_gpu):
    """configs[3] at its FULL size (bs=16, 64x64 latents, 5 tokens, SD-v1.5-sized UNet, LoRA r=8) through a size-independent property: every
    loss term is a batch mean, so a batch of 16 copies of one sample must give the loss and the gradients of that sample alone (B=1, itself
    checked against the fp32 oracle in test_full_size_training_gradients_match_oracle_autograd).  Covers what no oracle run can reach in
    test time: the M = 65536 GEMM / conv / weight-gradient shapes, their split-K choices and the 16-sample attention launches of the backward."""
    from photoverse_amd.adapters import PhotoVerseAdapter
    from photoverse_amd.clip import CLIPTextModel
    from photoverse_amd.lora import LoraConfig, LoRALinear, inject_adapter_in_model
    from photoverse_amd.train import TrainStep
    from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter
    torch.manual_seed(0)
    E, T, D = 5, 257, 1024
    unet = UNet2DConditionModel()
    set_visual_cross_attention_adapter(unet, (E,))
    inject_adapter_in_model(LoraConfig(r=8, lora_alpha=1), unet)
    g = torch.Generator().manual_seed(51)
    for m in unet.modules():
        if isinstance(m, LoRALinear):
            m.lora_B["default"].weight.data.normal_(0, 0.05, generator=g)
    text_encoder, image_adapter, text_adapter = CLIPTextModel(), PhotoVerseAdapter(D, 768, E), PhotoVerseAdapter(D, 768, E)
    for m in (unet, text_encoder, image_adapter, text_adapter):
        m.to("cuda")
    one = dict(noisy_latents=torch.randn(1, 4, 64, 64, generator=g), noise=torch.randn(1, 4, 64, 64, generator=g), timesteps=torch.tensor([633]),
               text_input_ids=torch.randint(0, 49000, (1, 77), generator=g), placeholder_idx=torch.tensor([[4]]),
               image_embeddings=[torch.randn(1, T, D, generator=g).half() for _ in range(E)])
    forced = [0.5] * 16
    forced[2], forced[11] = 0.1, 0.9

    def run(B):
        ts = TrainStep(unet, text_encoder, text_adapter, image_adapter, batch=B, h=64, w=64, n_tokens=E, grad_scale=4096.0, fusion_seed=5)
        rep = lambda t: t.repeat(B, *([1] * (t.dim() - 1)))
        kw = {k: ([rep(e).cuda() for e in v] if isinstance(v, list) else (rep(v) if k == "timesteps" else rep(v).cuda())) for k, v in one.items()}
        out = ts.step(**kw, forced_fusion=forced)
        torch.cuda.synchronize()
        groups = ts.trainable_parameters()
        res = {k: torch.cat([p.grad.detach().float().flatten() for p in ps]).cpu() for k, ps in groups.items()}
        loss = float(out["loss"])
        del ts
        torch.cuda.empty_cache()
        return loss, res
    loss1, g1 = run(1)
    loss16, g16 = run(16)
    assert loss16 == pytest.approx(loss1, rel=2e-3)
    errs = {k: rel_l2(g16[k], g1[k]) for k in g1}
    print("bs=16 replicated vs bs=1 gradient rel-L2 per module:", errs)
    # measured: UNet (to_k_ip / to_v_ip, LoRA) 3.5e-4, image adapter 7.7e-4, text adapter 1.2e-3
    assert max(errs.values()) < 3e-3, errs
