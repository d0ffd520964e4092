"""GPU parity of the HIP VAE decoder (infer.py:121-123) and encoder (infer.py:63) vs the CPU oracle."""
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
    # a decoder-only model ignores the encoder.* / quant_conv.* keys of an HF checkpoint
    from photoverse_amd.vae import AutoencoderKL
    sd = dict(ref.state_dict())
    sd["encoder.conv_in.weight"] = torch.zeros(1)
    sd["quant_conv.weight"] = torch.zeros(1)
    AutoencoderKL(**TINY, with_encoder=False).load_state_dict(sd)


@pytest.fixture(scope="module")
def enc_pair():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from oracle.vae_ref import AutoencoderKLDecoderRef
    from photoverse_amd.vae import AutoencoderKL
    torch.manual_seed(5)
    ref = AutoencoderKLDecoderRef(**TINY, with_encoder=True).eval()
    hip = AutoencoderKL(**TINY)
    hip.load_state_dict(ref.state_dict())
    hip.to("cuda")
    return ref, hip


@pytest.mark.parametrize("B,hw", [(1, 32), (3, 32), (2, 64)])
def test_vae_encode_matches_oracle(enc_pair, B, hw):
    """vae.encode(x).latent_dist (infer.py:63): mean / clamped logvar / a sample drawn with the same eps."""
    ref, hip = enc_pair
    g = torch.Generator().manual_seed(B * 100 + hw)
    x = torch.rand(B, 3, hw, hw, generator=g) * 2 - 1
    eps = torch.randn(B, 4, hw // 2, hw // 2, generator=g)
    with torch.no_grad():
        exp = ref.encode(x).latent_dist
        got = hip.encode(x.cuda()).latent_dist
    assert got.mean.shape == exp.mean.shape == (B, 4, hw // 2, hw // 2)
    assert rel_l2(got.mean, exp.mean) < 5e-3
    assert rel_l2(got.logvar, exp.logvar) < 5e-3
    assert rel_l2(got.mean + got.std * eps.cuda(), exp.sample(eps=eps)) < 5e-3
    assert torch.equal(got.mode(), got.mean)
    torch.manual_seed(11)
    s1 = got.sample()
    torch.manual_seed(11)
    assert torch.equal(s1, got.sample()) and s1.shape == got.mean.shape
    # the stride-2 / pad-(0,1,0,1) downsample reads one zero row / column past the bottom / right edge only
    x2 = x.clone()
    x2[:, :, 0, :] = 0.5
    with torch.no_grad():
        assert rel_l2(hip.encode(x2.cuda()).latent_dist.mean, ref.encode(x2).latent_dist.mean) < 5e-3


def test_run_inference_from_noised_image(enc_pair):
    """infer.py:62-65: latents = add_noise(vae.encode(pixel_values).latent_dist.sample() * scaling_factor, noise, t_0)."""
    from photoverse_amd.infer import run_inference
    from photoverse_amd.modeling_utils import load_models
    from oracle.unet_ref import TINY_CONFIG
    _, hip_vae = enc_pair
    vis = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2, image_size=56, patch_size=14)
    txt = dict(vocab_size=49408, hidden_size=768, num_attention_heads=12, intermediate_size=512, num_hidden_layers=1)
    tok, te, vae, unet, ie, ia, ta, sch, _ = load_models(None, 1, unet_config=TINY_CONFIG, vision_config=vis, text_config=txt, seed=3)
    for m in (unet, te, ie, ia, ta):
        m.to("cuda")
    g = torch.Generator().manual_seed(4)
    ex = {"pixel_values": torch.rand(2, 3, 32, 32, generator=g) * 2 - 1, "pixel_values_clip": torch.randn(2, 3, 56, 56, generator=g),
          "text_input_ids": torch.randint(0, 1000, (2, 77), generator=g), "concept_placeholder_idx": torch.tensor([[5], [3]])}
    kw = dict(latent_size=16, guidance_scale=3.0, timesteps=2, seed=1)
    with torch.no_grad():
        a = run_inference(ex, tok, ie, te, unet, ta, ia, hip_vae, sch, "cuda", [1], from_noised_image=True, **kw)
        b = run_inference(ex, tok, ie, te, unet, ta, ia, hip_vae, sch, "cuda", [1], from_noised_image=True, **kw)
        c = run_inference(ex, tok, ie, te, unet, ta, ia, hip_vae, sch, "cuda", [1], from_noised_image=False, **kw)
    assert a.shape == (2, 3, 32, 32) and torch.isfinite(a).all()
    assert torch.equal(a, b)                      # same seed -> same posterior sample and noise
    assert not torch.allclose(a, c)               # the start is the noised image, not pure noise


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


@pytest.mark.gpu
def test_decoder_data_gradient_on_the_training_tape():
    """vae_train.decode_on_tape: forward = AutoencoderKL.decode + clamp(-1, 1) (infer.py:121-123), backward = d<R, image>/dz against torch
    autograd over the oracle decoder (GroupNorm / conv / GEMM-composed mid-block attention / upsampling / conv_out / clamp mask)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")
    from oracle.vae_ref import AutoencoderKLDecoderRef
    from photoverse_amd.tape import Tape
    from photoverse_amd.vae import AutoencoderKL
    from photoverse_amd.vae_train import decode_on_tape
    cfg = dict(block_out_channels=(128, 128, 256, 256), layers_per_block=1)
    torch.manual_seed(3)
    ref = AutoencoderKLDecoderRef(**cfg).eval()
    vae = AutoencoderKL(**cfg)
    vae.load_state_dict(ref.state_dict(), strict=False)
    vae.to("cuda")
    g = torch.Generator().manual_seed(4)
    B, h = 2, 16
    z = torch.randn(B, 4, h, h, generator=g) * 3.0          # large latents: part of the image saturates the clamp
    R = torch.randn(B, 3, 8 * h, 8 * h, generator=g)
    S = 256.0
    tp = Tape("cuda", S)
    zbuf = tp.rf.hold(z.cuda())
    dec = decode_on_tape(tp, vae, zbuf)
    seed = tp.rb.hold((R * S).cuda())

    def set_seed():
        dec.dimg.g = seed
    tp.back.append(set_seed)
    tp.build_backward()
    tp.rf.run()
    tp.rb.run()
    torch.cuda.synchronize()
    zr = z.clone().requires_grad_()
    img = ref.decode(zr).sample.clamp(-1, 1)
    (img * R).sum().backward()
    sat = (img.detach().abs() >= 1).float().mean().item()
    err_img, err_g = rel_l2(dec.img, img.detach()), rel_l2(dec.dz.g / S, zr.grad)
    print(f"decoder on tape: image rel-L2 {err_img:.3e}, dz rel-L2 {err_g:.3e}, clamped fraction {sat:.3f}")
    assert 0.01 < sat < 0.9
    assert err_img < 5e-3
    assert err_g < 2e-2
