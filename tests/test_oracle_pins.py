"""CPU: pin the oracle against everything that can pin it in this image
(SURVEY.md section 8c): the real reference adapter (golden vectors), the
clip.py worked example, installed transformers CLIP, and torch SDPA."""
import os

import pytest
import torch
import torch.nn.functional as F

from oracle.adapters_ref import PhotoVerseAdapterRef
from oracle.clip_ref import CLIPTextModelRef, CLIPVisionModelRef, inject_concept_embeddings_ref
from oracle.unet_ref import (AttentionRef, PhotoVerseAttnProcessor2_0Ref, TINY_CONFIG, UNet2DConditionModelRef,
                             set_visual_cross_attention_adapter_ref, get_visual_cross_attention_values_norm_ref)


def test_adapter_matches_real_reference(golden_dir):
    g = torch.load(os.path.join(golden_dir, "adapter_golden.pt"))
    torch.manual_seed(g["weights_seed"])
    ad = PhotoVerseAdapterRef(1024, 768, num_tokens=2).eval()
    sd = ad.state_dict()
    assert set(sd) == set(g["weight_checksums"])            # same state-dict names as models/adapters.py
    for k, (s1, s2, head) in g["weight_checksums"].items():
        assert sd[k].double().sum().item() == pytest.approx(s1, rel=1e-12, abs=1e-12)
        assert (sd[k].double() ** 2).sum().item() == pytest.approx(s2, rel=1e-12)
        assert torch.equal(sd[k].flatten()[:4], head)
    embs = [e.float() for e in g["embs"]]
    with torch.no_grad():
        for key, ti in (("none", None), ("full", "full"), ("0", 0), ("1", 1)):
            out = ad(embs, token_index=ti)
            assert out.shape == g["outs"][key].shape
            torch.testing.assert_close(out, g["outs"][key], rtol=0, atol=0)
    # adapters.py:32-37: integer token_index returns (B,1,768) == column of the full output
    assert torch.equal(g["outs"]["0"], g["outs"]["full"][:, :1])


def test_inject_matches_clip_py_worked_example(golden_dir):
    g = torch.load(os.path.join(golden_dir, "inject_golden.pt"))
    out = inject_concept_embeddings_ref(g["old"], g["concept"], g["idx"])
    assert torch.equal(out, g["expected"])
    # clip.py:21-23 literal: new[10:] = old[6:73]; new[5:10] = concept
    assert torch.equal(out[0, 10:], g["old"][0, 6:73]) and torch.equal(out[0, 5:10], g["concept"][0])
    out1 = inject_concept_embeddings_ref(g["old"], g["one"], g["idx"].view(-1))   # (B,) index form, single token
    assert torch.equal(out1, g["expected_one"])


def test_clip_vision_matches_installed_transformers():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    torch.manual_seed(0)
    cfg = CLIPVisionConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2, image_size=56,
                           patch_size=14, hidden_act="quick_gelu")
    hf = CLIPVisionModel(cfg).eval()
    ref = CLIPVisionModelRef(128, 2, 256, 3, 56, 14).eval()
    sd = {(k if k.startswith("vision_model.") else "vision_model." + k): v for k, v in hf.state_dict().items()}
    ref.load_state_dict(sd)
    x = torch.randn(2, 3, 56, 56)
    with torch.no_grad():
        a = hf(x, output_hidden_states=True)
        b = ref(x)
    torch.testing.assert_close(b[0], a[0], rtol=1e-5, atol=1e-5)
    assert len(b[2]) == len(a.hidden_states) == 4
    for i in range(4):
        torch.testing.assert_close(b[2][i], a.hidden_states[i], rtol=1e-5, atol=1e-5)
    assert torch.equal(b[0], b[2][-1])                       # [0] is the last hidden state, no post-LN (infer.py:80)


def test_clip_text_matches_installed_transformers_and_injects():
    from transformers import CLIPTextConfig, CLIPTextModel
    torch.manual_seed(0)
    cfg = CLIPTextConfig(vocab_size=500, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                         max_position_embeddings=77, hidden_act="quick_gelu", bos_token_id=1, eos_token_id=2)
    hf = CLIPTextModel(cfg).eval()
    ref = CLIPTextModelRef(500, 64, 2, 128, 2, 77).eval()
    sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in hf.state_dict().items()}
    ref.load_state_dict(sd)
    ids = torch.randint(0, 500, (2, 77))
    with torch.no_grad():
        a = hf(input_ids=ids)[0]
        b = ref({"text_input_ids": ids})[0]
        torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-5)
        # injected forward == stock forward on the pre-injected embeddings
        concept = torch.randn(2, 1, 64)
        idx = torch.tensor([[4], [9]])
        c = ref({"text_input_ids": ids, "concept_text_embeddings": concept, "concept_placeholder_idx": idx})[0]
        emb = hf.get_input_embeddings()(ids)
        emb[0, 4], emb[1, 9] = concept[0, 0], concept[1, 0]
        # causal: positions before the placeholder are unchanged
        torch.testing.assert_close(c[0, :4], b[0, :4], rtol=1e-5, atol=1e-5)
        assert not torch.allclose(c[0, 4:], b[0, 4:])
    with pytest.raises(ValueError):
        ref(None)


def test_photoverse_processor_math_vs_sdpa():
    """attention_processor.py:297-423 by hand: SDPA(q,kt,vt) + SDPA(q,kip,vip) -> to_out."""
    torch.manual_seed(0)
    attn = AttentionRef(320, 768, heads=8, dim_head=40)
    proc = PhotoVerseAttnProcessor2_0Ref(320, 768, num_tokens=(5,))
    attn.set_processor(proc)
    h, text, ip = torch.randn(2, 64, 320), torch.randn(2, 77, 768), torch.randn(2, 5, 768)

    def heads(x):
        return x.view(2, -1, 8, 40).transpose(1, 2)

    with torch.no_grad():
        out = attn(h, encoder_hidden_states=(text, ip))
        q = heads(attn.to_q(h))
        o = F.scaled_dot_product_attention(q, heads(attn.to_k(text)), heads(attn.to_v(text)))
        vip = heads(proc.to_v_ip[0](ip))
        oip = F.scaled_dot_product_attention(q, heads(proc.to_k_ip[0](ip)), vip)
        exp = attn.to_out[0]((o + oip).transpose(1, 2).reshape(2, 64, 320))
    torch.testing.assert_close(out, exp, rtol=1e-5, atol=1e-5)
    assert proc.to_v_ip_norm.shape == (2, 8, 5, 1)
    torch.testing.assert_close(proc.to_v_ip_norm, vip.norm(dim=-1, keepdim=True))
    # list form and deprecated bare-tensor form (:258-273)
    with torch.no_grad():
        torch.testing.assert_close(attn(h, encoder_hidden_states=(text, [ip])), out)
        torch.testing.assert_close(attn(h, encoder_hidden_states=torch.cat([text, ip], 1)), out)
    # grad-mode random fusion (:413-420), forced
    for seed, fn in ((0.1, lambda: 2.0 * o), (0.9, lambda: 2.0 * oip), (0.5, lambda: o + oip)):
        proc.forced_fusion_seed = seed
        got = attn(h, encoder_hidden_states=(text, ip))
        exp = attn.to_out[0](fn().transpose(1, 2).reshape(2, 64, 320))
        torch.testing.assert_close(got, exp, rtol=1e-5, atol=1e-5)


def test_processor_init_errors():
    with pytest.raises(ValueError):
        PhotoVerseAttnProcessor2_0Ref(320, 768, fusion_rules=(0.5, 0.6))
    with pytest.raises(ValueError):
        PhotoVerseAttnProcessor2_0Ref(320, 768, fusion_rules=[1 / 3, 2 / 3])
    with pytest.raises(ValueError):
        PhotoVerseAttnProcessor2_0Ref(320, 768, num_tokens=(5,), scale=[1.0, 2.0])


def test_unet_structure_and_regression(golden_dir):
    with torch.device("meta"):
        full = UNet2DConditionModelRef()
    assert sum(p.numel() for p in full.parameters()) == 859_520_964       # public SD-v1.5 UNet parameter count
    set_visual_cross_attention_adapter_ref(full, (5,))
    procs = full.attn_processors
    assert len(procs) == 32 and sum(isinstance(p, PhotoVerseAttnProcessor2_0Ref) for p in procs.values()) == 16
    ip_params = sum(p.numel() for n, p in full.named_parameters() if "processor" in n)
    assert ip_params == 2 * 768 * (5 * 320 + 5 * 640 + 6 * 1280)           # 19.17 M
    keys = [k for k in full.state_dict() if "attn2" in k and any(s in k for s in ("processor", "to_q", "to_k", "to_v"))]
    assert len(keys) == 16 * 5                                             # save_progress filter, modeling_utils.py:34-37

    g = torch.load(os.path.join(golden_dir, "tiny_unet_golden.pt"))
    from oracle.infer_ref import denoise_ref, draw_noise_ref
    torch.manual_seed(g["weights_seed"])
    unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_visual_cross_attention_adapter_ref(unet, (5,))
    gen = torch.Generator().manual_seed(g["cond_seed"])
    text, utext = torch.randn(2, 77, 768, generator=gen), torch.randn(2, 77, 768, generator=gen)
    ip, uip = torch.randn(2, 1, 768, generator=gen), torch.randn(2, 1, 768, generator=gen)
    noise = draw_noise_ref(2, 4, 16, seed=g["noise_seed"])
    with torch.no_grad():
        eps = unet(noise, torch.tensor(500), encoder_hidden_states=(text, ip)).sample
    torch.testing.assert_close(eps, g["eps_t500"], rtol=1e-4, atol=1e-5)
    assert get_visual_cross_attention_values_norm_ref(unet).shape == (2, 4 * 8 * 1)   # 4 cross-attn layers x 8 heads x P=1
    lat = denoise_ref(unet, noise, (text, ip), (utext, uip), guidance_scale=7.5, timesteps=2)
    torch.testing.assert_close(lat, g["latents_2step"], rtol=1e-4, atol=1e-4)


def test_scheduler_tables():
    from oracle.scheduler_ref import DPMSolverMultistepRef
    s = DPMSolverMultistepRef()
    s.set_timesteps(50)
    assert s.timesteps[0].item() == 951 and s.timesteps[-1].item() == 20 and len(s.timesteps) == 50   # leading, offset 1
    assert s.sigmas[-1] == 0.0 and len(s.sigmas) == 51
    # exactness property: if the model predicts the true noise of x_t = a*x0 + s*eps, the solver lands on x0
    x0, eps = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    x = x0 + float(s.sigmas[0]) * eps          # sigma-space sample (alpha folded), then alpha-scale
    a0 = 1.0 / (float(s.sigmas[0]) ** 2 + 1) ** 0.5
    x = x * a0
    for i, t in enumerate(s.timesteps):
        x = s.step(eps, t, x)
    torch.testing.assert_close(x, x0, rtol=1e-3, atol=1e-3)


def test_samplers_converge_to_the_exact_probability_flow_solution():
    """Known-answer test for the UNPINNED samplers (diffusers is not installable): for Gaussian data x0 ~ N(0, s^2 I) the optimal
    epsilon-predictor is linear, eps*(x_t) = sigma_t x_t / (alpha_t^2 s^2 + sigma_t^2), and the probability-flow ODE it drives has
    the closed form x_t = x_T sqrt(var_t / var_T).
    (1) On a grid uniform in lambda = log(alpha/sigma) the DPM-Solver++(2M) update must converge to it at SECOND order - this pins
        the data-prediction conversion and the multistep coefficients.
    (2) On the reference's own grid ("leading" spacing, uniform in t, final jump to sigma = 0) the last steps have h = O(1) in
        lambda whatever the step count (lambda ~ -1/2 log t), so the error only falls like 1/n - checked as monotone decrease, which
        still pins the noise schedule and the timestep / sigma tables (a wrong table gives an error that does not shrink).
    (3) DDIM: first order on its own grid."""
    import numpy as np
    from oracle.scheduler_ref import DDIMRef, DPMSolverMultistepRef
    s2 = 0.25
    x_init = torch.randn(64, generator=torch.Generator().manual_seed(0), dtype=torch.float64)
    rel = lambda x, exact: ((x - exact).norm() / exact.norm()).item()

    def dpm(n, lam_grid):
        sch = DPMSolverMultistepRef()
        sch.set_timesteps(n)
        if lam_grid:
            sch.sigmas = np.exp(-np.linspace(-2.5, 3.0, n + 1))          # sigma/alpha = exp(-lambda); stops at lambda = 3 (no jump to 0)
        var = lambda i: (lambda a, sg: a * a * s2 + sg * sg)(*sch._alpha_sigma(np.float64(sch.sigmas[i])))
        x = x_init.clone()
        for i, t in enumerate(sch.timesteps):
            a, sg = sch._alpha_sigma(np.float64(sch.sigmas[i]))
            x = sch.step(float(sg / (a * a * s2 + sg * sg)) * x, t, x)
        return rel(x, x_init * float(np.sqrt(var(n) / var(0))))

    e = {n: dpm(n, True) for n in (40, 80, 160)}
    assert e[160] < 1e-4, e
    assert 3.5 < e[40] / e[80] < 4.5 and 3.5 < e[80] / e[160] < 4.5, e          # second order: h/2 -> error/4
    f = {n: dpm(n, False) for n in (20, 40, 80, 160)}
    assert f[160] < 1.5e-2 and all(1.6 < f[n] / f[2 * n] < 2.4 for n in (20, 40, 80)), f

    def ddim(n):
        sch = DDIMRef()
        sch.set_timesteps(n)
        acp = sch.alphas_cumprod
        var = lambda a: float(a) * s2 + (1.0 - float(a))
        x = x_init.clone()
        for t in sch.timesteps:
            a = acp[int(t)]
            x = sch.step(float((1 - a).sqrt() / var(a)) * x, t, x)
        return rel(x, x_init * float(np.sqrt(var(acp[0]) / var(acp[int(sch.timesteps[0])]))))

    d = {n: ddim(n) for n in (20, 40, 80, 160)}
    assert d[160] < 2e-2 and all(1.5 < d[n] / d[2 * n] < 2.6 for n in (20, 40, 80)), d
