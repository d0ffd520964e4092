"""GPU parity of the HIP VAE decoder (infer.py:121-123) vs the CPU oracle."""
import pytest
import torch

pytestmark = pytest.mark.gpu
TINY = dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256), layers_per_block=1, norm_num_groups=32, scaling_factor=0.18215)


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


@pytest.fixture(scope="module")
def pair():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from oracle.vae_ref import AutoencoderKLDecoderRef
    from photoverse_amd.vae import AutoencoderKL
    torch.manual_seed(0)
    ref = AutoencoderKLDecoderRef(**TINY).eval()
    hip = AutoencoderKL(**TINY)
    hip.load_state_dict(ref.state_dict())
    hip.to("cuda")
    return ref, hip


@pytest.mark.parametrize("B,hw", [(1, 16), (3, 16), (2, 32)])
def test_vae_decode_matches_oracle(pair, B, hw):
    ref, hip = pair
    z = torch.randn(B, 4, hw, hw, generator=torch.Generator().manual_seed(B + hw))
    with torch.no_grad():
        exp = ref.decode(z).sample
        got = hip.decode(z.cuda()).sample
    assert got.shape == exp.shape == (B, 3, 2 * hw, 2 * hw)
    assert rel_l2(got, exp) < 5e-3


def test_vae_sub_batching_and_extra_keys(pair):
    ref, hip = pair
    hip.MAX_OPERAND_BYTES = 256 * 32 * 32 * 2 * 2          # forces sub-batches of 2 images
    z = torch.randn(5, 4, 16, 16, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        exp = ref.decode(z).sample
        got = hip.decode(z.cuda()).sample
    assert rel_l2(got, exp) < 5e-3
    type(hip).MAX_OPERAND_BYTES = 1 << 30
    del hip.MAX_OPERAND_BYTES
    # an HF checkpoint also carries encoder.* / quant_conv.* keys: ignored on load
    sd = dict(ref.state_dict())
    sd["encoder.conv_in.weight"] = torch.zeros(1)
    sd["quant_conv.weight"] = torch.zeros(1)
    hip.load_state_dict(sd)


def test_run_inference_returns_clamped_images_with_vae(pair):
    """infer.py:121-123: images = vae.decode(latents / scaling_factor).sample.clamp(-1, 1)."""
    from photoverse_amd.infer import run_inference
    from photoverse_amd.modeling_utils import load_models
    from oracle.unet_ref import TINY_CONFIG
    _, hip_vae = pair
    vis = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2, image_size=56, patch_size=14)
    txt = dict(vocab_size=49408, hidden_size=768, num_attention_heads=12, intermediate_size=512, num_hidden_layers=1)
    tok, te, vae, unet, ie, ia, ta, sch, _ = load_models(None, 1, unet_config=TINY_CONFIG, vision_config=vis, text_config=txt, seed=3)
    for m in (unet, te, ie, ia, ta):
        m.to("cuda")
    g = torch.Generator().manual_seed(4)
    ex = {"pixel_values": torch.zeros(2, 3, 128, 128), "pixel_values_clip": torch.randn(2, 3, 56, 56, generator=g),
          "text_input_ids": torch.randint(0, 1000, (2, 77), generator=g), "concept_placeholder_idx": torch.tensor([[5], [3]])}
    with torch.no_grad():
        lat = run_inference(ex, tok, ie, te, unet, ta, ia, None, sch, "cuda", [1], latent_size=16, guidance_scale=3.0, timesteps=2, seed=1)
        img = run_inference(ex, tok, ie, te, unet, ta, ia, hip_vae, sch, "cuda", [1], latent_size=16, guidance_scale=3.0, timesteps=2, seed=1)
        exp = hip_vae.decode(lat / hip_vae.config.scaling_factor).sample.clamp(-1, 1)
    assert img.shape == (2, 3, 32, 32) and img.min() >= -1 and img.max() <= 1
    assert torch.equal(img, exp)


def test_full_size_vae_decode_matches_oracle():
    """SD-v1.5 VAE decoder at its real size (49.5 M params): one 64x64 latent -> 512x512 image vs the fp32 CPU oracle."""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from oracle.vae_ref import AutoencoderKLDecoderRef
    from photoverse_amd.vae import AutoencoderKL
    torch.manual_seed(1)
    ref = AutoencoderKLDecoderRef().eval()
    assert sum(p.numel() for p in ref.parameters()) == 49_490_199          # public SD VAE decoder (+ post_quant_conv) size
    hip = AutoencoderKL()
    hip.load_state_dict(ref.state_dict())
    hip.to("cuda")
    z = torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        exp = ref.decode(z).sample
        got = hip.decode(z.cuda()).sample
    err = rel_l2(got, exp)
    print(f"full-size VAE decode rel-L2 vs fp32 oracle: {err:.3e}")
    assert got.shape == (1, 3, 512, 512) and err < 5e-3
