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
        assert rel_l2(got, exp) < 1e-2
    with pytest.raises(NotImplementedError):
        run_inference(example, tok, image_encoder, text_encoder, unet, text_adapter, image_adapter, None, scheduler, "cuda", [1],
                      latent_size=16, timesteps=2, from_noised_image=True)
