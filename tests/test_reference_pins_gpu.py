"""GPU: the HIP path against fixtures produced by EXECUTING THE REFERENCE'S OWN CODE in the build container
(``oracle/make_ref_golden.py``; CPU counterparts in ``tests/test_reference_pins.py``).  No oracle arithmetic is involved here: the
expected tensors come from /root/reference's function bodies, the inputs and seeds from the fixture, the weights from
``oracle.seeded.fill_state_`` (a data generator, checked against the fixture's per-tensor checksums)."""
import os
import warnings

import pytest
import torch

from oracle.seeded import checksums, fill_state_

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a HIP device")


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def _check_sums(module, sums, prefix=""):
    got = checksums(module)
    for k, (s1, s2) in sums.items():
        assert got[prefix + k][0] == pytest.approx(s1, rel=1e-9, abs=1e-9) and got[prefix + k][1] == pytest.approx(s2, rel=1e-9), k


def test_text_forward_with_injection_matches_reference_code(need_gpu, golden_dir):
    """models/clip.py:17-102 (executed) vs ``pv_clip_text_embed`` + the causal encoder: E = 0 / 1 / 5 concept tokens."""
    from photoverse_amd.clip import CLIPTextModel, patch_clip_text_transformer
    g = _load(golden_dir, "ref_text_golden.pt")
    c = g["config"]
    hip = patch_clip_text_transformer(CLIPTextModel(vocab_size=c["vocab_size"], hidden_size=c["hidden_size"],
                                                    num_attention_heads=c["num_attention_heads"], intermediate_size=c["intermediate_size"],
                                                    num_hidden_layers=c["num_hidden_layers"], max_position_embeddings=c["max_position_embeddings"]))
    sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in g["state_dict"].items() if "position_ids" not in k}
    hip.load_state_dict(sd)
    hip.to("cuda")
    ids = g["ids"].cuda()
    with torch.no_grad():
        for E, o in g["outs"].items():
            d = {"text_input_ids": ids}
            if E:
                d.update(concept_text_embeddings=o["concept"].cuda(), concept_placeholder_idx=o["idx"].cuda())
            got = hip(d)[0]
            err = rel_l2(got, o["last_hidden_state"])
            print(f"text forward E={E}: rel-L2 vs reference code {err:.3e}")
            assert err < 3e-3
    with pytest.raises(ValueError, match=g["none_error"]):
        hip(None)


@pytest.mark.parametrize("P", [1, 5])
def test_processor_matches_reference_call(need_gpu, golden_dir, P):
    """models/attention_processor.py:245-435 (executed, with its own ``torch.rand(1).item()`` draw) vs the HIP processor: no_grad sum in
    the tuple / list / bare-tensor conventions, ``to_v_ip_norm``, and the three grad-mode fusion branches with gradients w.r.t.
    hidden states, text, image tokens, to_k_ip and to_v_ip."""
    from photoverse_amd.attention_processor import PhotoVerseAttnProcessor2_0
    from photoverse_amd.unet import Attention
    g = _load(golden_dir, "ref_processor_golden.pt")
    C, heads = g["C"], g["heads"]
    attn = Attention(C, cross_attention_dim=768, heads=heads, dim_head=C // heads)
    fill_state_(attn, g["attn_seed"])
    _check_sums(attn, g["attn_checksums"])
    proc = PhotoVerseAttnProcessor2_0(hidden_size=C, cross_attention_dim=768, num_tokens=(P,))
    fill_state_(proc, g["proc_seed"])
    if P == 1:
        _check_sums(proc, g["proc_checksums"])
    attn.set_processor(proc)
    attn.to("cuda")
    c = g["cases"][P]
    hs, text, ip, G = (c[k].float().cuda() for k in ("hs", "text", "ip", "G"))
    with torch.no_grad():
        got = attn(hs, encoder_hidden_states=(text, ip))
        e = rel_l2(got, c["nograd_tuple"])
        print(f"processor P={P} no_grad: {e:.3e}; vnorm {rel_l2(proc.to_v_ip_norm, c['vnorm']):.3e}")
        assert e < 2e-3
        assert proc.to_v_ip_norm.shape == c["vnorm"].shape and rel_l2(proc.to_v_ip_norm, c["vnorm"]) < 1e-3
        assert torch.equal(attn(hs, encoder_hidden_states=(text, [ip])), got)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            assert rel_l2(attn(hs, encoder_hidden_states=torch.cat([text, ip], 1)), c["nograd_tensor"]) < 2e-3
    for region in ("text", "sum", "ip"):
        ex = c["grad_" + region]
        ps = [proc.to_k_ip[0].weight, proc.to_v_ip[0].weight]
        for p_ in ps:
            p_.requires_grad_(True)
            p_.grad = None
        h, t, i = (v.clone().requires_grad_(True) for v in (hs, text, ip))
        torch.manual_seed(ex["torch_seed"])              # the product draws torch.rand(1).item() on the CPU generator like :414
        with torch.enable_grad():
            o = attn(h, encoder_hidden_states=(t, i))
            loss = (o.float() * G).sum() + 0.3 * proc.to_v_ip_norm.sum()
        loss.backward()

        def gr(v):
            return torch.zeros_like(v) if v.grad is None else v.grad.float()
        errs = {"out": rel_l2(o.detach(), ex["out"]), "d_ip": rel_l2(gr(i), ex["d_ip"]), "d_to_v_ip": rel_l2(gr(ps[1])[::4, ::4], ex["d_to_v_ip"])}
        if P == 1 and region == "ip":                     # one image token, text branch dropped: the output does not depend on the query
            assert float(ex["d_hs"].norm()) < 1e-4 * float(ex["d_ip"].norm()) and float(gr(h).norm()) < 1e-3 * float(gr(i).norm())
        else:
            errs["d_hs"] = rel_l2(gr(h), ex["d_hs"])
        if region != "ip":                                # u > 2/3 drops the text branch: its gradient is exactly zero on both sides
            errs["d_text"] = rel_l2(gr(t)[:, :, ::8], ex["d_text"])
        else:
            assert float(gr(t).abs().max()) == 0.0 and float(ex["d_text"].abs().max()) == 0.0
        if P == 1:                                        # softmax over ONE image token is 1 whatever its key: d to_k_ip is rounding noise
            assert float(ex["d_to_k_ip"].norm()) < 1e-5 * float(ex["d_to_v_ip"].norm())
            assert float(gr(ps[0]).norm()) < 1e-3 * float(gr(ps[1]).norm())
        elif region != "text":
            errs["d_to_k_ip"] = rel_l2(gr(ps[0])[::4, ::4], ex["d_to_k_ip"])
        else:
            assert float(gr(ps[0]).abs().max()) == 0.0 and float(ex["d_to_k_ip"].abs().max()) == 0.0
        print(f"processor P={P} grad-mode region {region} (u = {ex['u']:.3f}):", {k: f"{v:.2e}" for k, v in errs.items()})
        assert errs["out"] < 2e-3 and max(errs.values()) < 5e-3, errs


def test_unet_install_and_vnorm_match_reference_functions(need_gpu, golden_dir):
    """models/unet.py:8-47 executed on the tiny UNet with the REFERENCE processor class installed vs the HIP UNet with the product's
    ``set_visual_cross_attention_adapter`` / ``get_visual_cross_attention_values_norm``."""
    from oracle.unet_ref import TINY_CONFIG
    from photoverse_amd.unet import UNet2DConditionModel, get_visual_cross_attention_values_norm, set_visual_cross_attention_adapter
    g = _load(golden_dir, "ref_unet_golden.pt")
    hip = UNet2DConditionModel(**TINY_CONFIG)
    set_visual_cross_attention_adapter(hip, (5,))
    fill_state_(hip, g["weights_seed"])
    _check_sums(hip, g["checksums"])
    hip.repack()
    hip.to("cuda")
    inv = {n: (getattr(p, "hidden_size", None), getattr(p, "cross_attention_dim", None)) for n, p in hip.attn_processors.items()}
    assert inv == {n: (h, c) for n, (_cls, h, c) in g["processors"].items()}
    with torch.no_grad():
        eps = hip(g["x"].cuda(), torch.tensor(g["t"]), encoder_hidden_states=(g["text"].cuda(), g["ip"].cuda())).sample
        vn = get_visual_cross_attention_values_norm(hip)
    print(f"tiny UNet with reference-installed processors: eps {rel_l2(eps, g['eps']):.3e}, vnorm {rel_l2(vn, g['vnorm']):.3e}")
    assert rel_l2(eps, g["eps"]) < 2.5e-3
    assert vn.shape == g["vnorm"].shape and rel_l2(vn, g["vnorm"]) < 1e-3


def test_arcface_loss_matches_reference_classes(need_gpu, golden_dir):
    """models/arcface_resnet.py:12-134 + models/loss.py:26-78 (executed, seeded weights) vs the HIP launch plan: loss for both targets and
    the image gradient (PReLU / max-pool kinks under an fp16 forward bound the gradient agreement, tests/test_loss_gpu.py)."""
    from photoverse_amd.loss import ArcFaceResNet18, FaceLoss
    g = _load(golden_dir, "ref_arcface_golden.pt")
    net = ArcFaceResNet18()
    fill_state_(net, g["weights_seed"])
    assert list(net.state_dict().keys()) == g["state_keys"]
    _check_sums(net, g["checksums"])
    fl = FaceLoss("cuda", "arcface", model=net)
    loss, dimg = fl.loss_and_grad(g["x"].cuda(), g["x_gen"].cuda())
    lmin = fl(g["x"].cuda(), g["x_gen"].cuda(), maximize=False)
    print(f"face loss {loss.item():.5f} vs reference code {g['loss'].item():.5f}; minimize {float(lmin):.5f} vs {g['loss_minimize'].item():.5f}; "
          f"d x_gen rel-L2 {rel_l2(dimg, g['d_x_gen']):.3e}")
    assert loss.item() == pytest.approx(g["loss"].item(), rel=3e-2, abs=5e-4)
    assert float(lmin) == pytest.approx(g["loss_minimize"].item(), rel=1e-3)
    assert rel_l2(dimg, g["d_x_gen"]) < 1e-1


def test_adapter_configs4_shape_matches_reference_class(need_gpu, golden_dir):
    """HIP adapter with 17 mapping pairs on 6 CLIP hidden states (BASELINE configs[4] conditioning: extra_num_tokens = 16, five encoder layers + the
    last) vs the REAL models/adapters.py class (fixture produced by executing it)."""
    from photoverse_amd.adapters import PhotoVerseAdapter
    g = _load(golden_dir, "ref_adapter17_golden.pt")
    ad = PhotoVerseAdapter(1024, 768, num_tokens=17)
    fill_state_(ad, g["weights_seed"])
    assert len(ad.state_dict()) == g["n_state"]
    ad.to("cuda")
    gen = torch.Generator().manual_seed(g["input_seed"])
    embs = [torch.randn(2, 257, 1024, generator=gen).half().cuda() for _ in range(6)]
    for key, ti in (("none", None), ("0", 0), ("5", 5)):
        out = ad(embs, token_index=ti)
        err = rel_l2(out, g["outs"][key])
        print(f"17-mapping adapter token_index={ti}: rel-L2 vs reference class {err:.3e}")
        assert out.shape == g["outs"][key].shape and err < 3e-3
