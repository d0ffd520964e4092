"""CPU: the oracle restatements against fixtures produced by EXECUTING THE REFERENCE'S OWN CODE (``oracle/make_ref_golden.py`` /
``oracle/ref_exec.py``: definitions loaded from the source text of /root/reference in the build container).  These pin SURVEY.md
section 8 rows A1 / A2 (processor), A5 / A6 (install + V-norm stack), A8 / A9 (injection, dict-input text forward) and the ArcFace
identity loss of f3 on reference code, not on a hand-made restatement.  The GPU counterparts (HIP path vs the same fixtures) live in
``tests/test_reference_pins_gpu.py``."""
import os

import pytest
import torch

from oracle.seeded import checksums, fill_state_


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def _check_sums(module, sums):
    got = checksums(module)
    assert set(got) == set(sums), set(got) ^ set(sums)          # same state-dict names as the reference class
    for k, (s1, s2) in sums.items():
        assert got[k][0] == pytest.approx(s1, rel=1e-9, abs=1e-9) and got[k][1] == pytest.approx(s2, rel=1e-9), k


def test_inject_matches_reference_function(golden_dir):
    from oracle.clip_ref import inject_concept_embeddings_ref
    g = _load(golden_dir, "ref_inject_golden.pt")
    assert len(g["cases"]) == 4
    for c in g["cases"]:
        assert torch.equal(inject_concept_embeddings_ref(c["old"], c["concept"], c["idx"]), c["expected"])


def test_text_forward_matches_reference_function(golden_dir):
    """clip.py:29-102 executed over transformers' own modules vs ``CLIPTextModelRef`` with the same weights."""
    from oracle.clip_ref import CLIPTextModelRef
    g = _load(golden_dir, "ref_text_golden.pt")
    c = g["config"]
    ref = CLIPTextModelRef(c["vocab_size"], c["hidden_size"], c["num_attention_heads"], c["intermediate_size"], c["num_hidden_layers"],
                           c["max_position_embeddings"]).eval()
    sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in g["state_dict"].items()
          if "position_ids" not in k}
    ref.load_state_dict(sd)
    with torch.no_grad():
        for E, o in g["outs"].items():
            d = {"text_input_ids": g["ids"]}
            if E:
                d.update(concept_text_embeddings=o["concept"], concept_placeholder_idx=o["idx"])
            got = ref(d)
            torch.testing.assert_close(got[0], o["last_hidden_state"], rtol=2e-5, atol=2e-5)
            torch.testing.assert_close(got[1], o["pooled"], rtol=2e-5, atol=2e-5)
    assert g["none_error"] == "You have to specify either input_ids"
    with pytest.raises(ValueError, match="You have to specify either input_ids"):
        ref(None)


def _processor_setup(g, P):
    from oracle.unet_ref import AttentionRef, PhotoVerseAttnProcessor2_0Ref
    C, heads = g["C"], g["heads"]
    attn = AttentionRef(C, cross_attention_dim=768, heads=heads, dim_head=C // heads).eval()
    fill_state_(attn, g["attn_seed"])
    proc = PhotoVerseAttnProcessor2_0Ref(hidden_size=C, cross_attention_dim=768, num_tokens=(P,))
    fill_state_(proc, g["proc_seed"])
    return attn, proc


@pytest.mark.parametrize("P", [1, 5])
def test_processor_matches_reference_call(golden_dir, P):
    """attention_processor.py:245-435 executed (tuple / list / bare tensor, no_grad sum, the three grad-mode fusion branches with
    the reference's own ``torch.rand(1).item()`` draw, gradients) vs ``PhotoVerseAttnProcessor2_0Ref``."""
    g = _load(golden_dir, "ref_processor_golden.pt")
    attn, proc = _processor_setup(g, P)
    _check_sums(attn, g["attn_checksums"])
    if P == 1:
        _check_sums(proc, g["proc_checksums"])
    c = g["cases"][P]
    hs, text, ip, G = (c[k].float() for k in ("hs", "text", "ip", "G"))
    with torch.no_grad():
        torch.testing.assert_close(proc(attn, hs, encoder_hidden_states=(text, ip)), c["nograd_tuple"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(proc.to_v_ip_norm, c["vnorm"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(proc(attn, hs, encoder_hidden_states=(text, [ip])), c["nograd_list"], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(proc(attn, hs, encoder_hidden_states=torch.cat([text, ip], 1)), c["nograd_tensor"], rtol=1e-5, atol=1e-5)
    assert torch.equal(c["nograd_tuple"], c["nograd_list"])
    for region in ("text", "sum", "ip"):
        e = c["grad_" + region]
        ps = [proc.to_k_ip[0].weight, proc.to_v_ip[0].weight]
        for p_ in ps:
            p_.requires_grad_(True)
            p_.grad = None
        h, t, i = (v.clone().requires_grad_(True) for v in (hs, text, ip))
        torch.manual_seed(e["torch_seed"])                      # the restatement draws torch.rand(1).item() exactly like :414
        with torch.enable_grad():
            o = proc(attn, h, encoder_hidden_states=(t, i))
            loss = (o * G).sum() + 0.3 * proc.to_v_ip_norm.sum()
        loss.backward()
        torch.testing.assert_close(o.detach(), e["out"], rtol=1e-5, atol=1e-5)

        def gr(v):
            return torch.zeros_like(v) if v.grad is None else v.grad
        for got, exp in ((gr(h), e["d_hs"]), (gr(t)[:, :, ::8], e["d_text"]), (gr(i), e["d_ip"]), (gr(ps[0])[::4, ::4], e["d_to_k_ip"]),
                         (gr(ps[1])[::4, ::4], e["d_to_v_ip"])):
            assert (got - exp).norm() <= 2e-5 * exp.norm() + 1e-6
    # the three regions really are different branches
    assert not torch.allclose(c["grad_text"]["out"], c["grad_sum"]["out"]) and not torch.allclose(c["grad_ip"]["out"], c["grad_sum"]["out"])


def test_processor_init_errors_match_reference(golden_dir):
    from oracle.unet_ref import PhotoVerseAttnProcessor2_0Ref
    from photoverse_amd.attention_processor import PhotoVerseAttnProcessor2_0
    g = _load(golden_dir, "ref_processor_golden.pt")
    kws = {"fusion_type": dict(fusion_rules=[1 / 3, 2 / 3]), "fusion_sum": dict(fusion_rules=(0.5, 0.6)), "scale_len": dict(scale=[1.0, 2.0])}
    for key, msg in g["init_errors"].items():
        assert msg is not None
        for cls in (PhotoVerseAttnProcessor2_0Ref, PhotoVerseAttnProcessor2_0):      # oracle AND the product's host mirror
            with pytest.raises(ValueError) as ei:
                cls(hidden_size=320, cross_attention_dim=768, num_tokens=(5,), **kws[key])
            assert str(ei.value) == msg


def test_unet_install_and_vnorm_match_reference_functions(golden_dir):
    """models/unet.py:8-47 executed on the oracle's tiny UNet with the REFERENCE processor class vs the restated helpers."""
    from oracle.unet_ref import (TINY_CONFIG, UNet2DConditionModelRef, get_visual_cross_attention_values_norm_ref,
                                 set_visual_cross_attention_adapter_ref)
    g = _load(golden_dir, "ref_unet_golden.pt")
    unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_visual_cross_attention_adapter_ref(unet, (5,))
    fill_state_(unet, g["weights_seed"])
    _check_sums(unet, g["checksums"])
    inv = {n: (getattr(p, "hidden_size", None), getattr(p, "cross_attention_dim", None)) for n, p in unet.attn_processors.items()}
    assert inv == {n: (h, c) for n, (_cls, h, c) in g["processors"].items()}
    assert {cls for _n, (cls, _h, _c) in g["processors"].items()} == {"AttnProcessor2_0Ref", "PhotoVerseAttnProcessor2_0"}
    with torch.no_grad():
        eps = unet(g["x"], torch.tensor(g["t"]), encoder_hidden_states=(g["text"], g["ip"])).sample
        vn = get_visual_cross_attention_values_norm_ref(unet)
    torch.testing.assert_close(eps, g["eps"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(vn, g["vnorm"], rtol=1e-6, atol=1e-6)
    assert vn.shape == g["vnorm"].shape


def test_arcface_and_face_loss_match_reference_classes(golden_dir):
    """models/arcface_resnet.py:12-134 + models/loss.py:26-78 executed vs ``ArcFaceResNet18Ref`` / ``FaceLossRef``."""
    from oracle.arcface_ref import FaceLossRef
    g = _load(golden_dir, "ref_arcface_golden.pt")
    fl = FaceLossRef()
    fill_state_(fl.model, g["weights_seed"])
    assert list(fl.model.state_dict().keys()) == g["state_keys"]
    _check_sums(fl.model, g["checksums"])
    with torch.no_grad():
        torch.testing.assert_close(fl.model(g["gray"]), g["embedding"], rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(fl.preprocess(g["x"]), g["preprocess"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(fl.preprocess(g["x"][:, :1], normalize=False), g["preprocess_raw_1ch"], rtol=1e-6, atol=1e-5)
        torch.testing.assert_close(fl(g["x"], g["x_gen"], maximize=False), g["loss_minimize"], rtol=1e-5, atol=1e-6)
    xg = g["x_gen"].clone().requires_grad_(True)
    loss = fl(g["x"], xg)
    loss.backward()
    torch.testing.assert_close(loss.detach(), g["loss"], rtol=1e-5, atol=1e-6)
    assert (xg.grad - g["d_x_gen"]).norm() <= 1e-4 * g["d_x_gen"].norm()


def test_lora_oracle_is_independent_of_the_product_and_agrees_with_it():
    """``oracle/lora_ref.py`` (peft's published forward, written without the product) vs ``photoverse_amd.lora``: same keys, same merged
    weight, same un-merged forward.  The product file is never imported by the oracle (checked on the source text)."""
    import inspect

    import oracle.lora_ref as lr
    from photoverse_amd.lora import LoraConfig, inject_adapter_in_model
    assert "photoverse_amd" not in inspect.getsource(lr).replace("photoverse_amd/lora.py", "")
    import torch.nn as nn

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.attn2 = nn.Module()
            self.attn2.to_q, self.attn2.to_k, self.attn2.to_v = nn.Linear(32, 32, bias=False), nn.Linear(48, 32, bias=False), nn.Linear(48, 32, bias=False)
            self.attn2.to_out = nn.ModuleList([nn.Linear(32, 32), nn.Dropout(0.0)])
            self.attn1 = nn.Module()
            self.attn1.to_q = nn.Linear(32, 32, bias=False)
    torch.manual_seed(0)
    a, b = Blk(), Blk()
    b.load_state_dict(a.state_dict())
    targets = ["attn2.to_q", "attn2.to_k", "attn2.to_v", "attn2.to_out.0"]
    lr.inject_adapter_in_model_ref(a, r=4, lora_alpha=8, target_modules=targets)
    inject_adapter_in_model(LoraConfig(r=4, lora_alpha=8, target_modules=targets), b)
    assert set(a.state_dict()) == set(b.state_dict()) and any(k.endswith("to_out.0.lora_B.default.weight") for k in a.state_dict())
    assert not isinstance(a.attn1.to_q, lr.LoraLinearRef)
    g = torch.Generator().manual_seed(1)
    for k, v in a.state_dict().items():
        if "lora_" in k:
            v.copy_(torch.randn(v.shape, generator=g) * 0.1)
    b.load_state_dict(a.state_dict())
    x = torch.randn(5, 32, generator=g)
    y = a.attn2.to_q(x)
    exp = x @ a.attn2.to_q.base_layer.weight.T + 2.0 * (x @ a.attn2.to_q.lora_A["default"].weight.T) @ a.attn2.to_q.lora_B["default"].weight.T
    torch.testing.assert_close(y, exp, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(torch.nn.functional.linear(x, b.attn2.to_q.weight), y, rtol=1e-5, atol=1e-5)   # product's merged weight


def test_adapter_configs4_shape_matches_reference_class(golden_dir):
    """The REAL models/adapters.py class with 17 mapping pairs on 6 CLIP hidden states (extra_num_tokens = 16, five encoder layers + the last:
    BASELINE configs[4]) vs ``PhotoVerseAdapterRef`` - seeded weights, inputs re-drawn from the seed."""
    from oracle.adapters_ref import PhotoVerseAdapterRef
    g = _load(golden_dir, "ref_adapter17_golden.pt")
    ad = PhotoVerseAdapterRef(1024, 768, 17).eval()
    fill_state_(ad, g["weights_seed"])
    assert len(ad.state_dict()) == g["n_state"]
    gen = torch.Generator().manual_seed(g["input_seed"])
    embs = [torch.randn(2, 257, 1024, generator=gen).half().float() for _ in range(6)]
    with torch.no_grad():
        for key, ti in (("none", None), ("0", 0), ("5", 5)):
            out = ad(embs, token_index=ti)
            assert out.shape == g["outs"][key].shape == ((2, 6, 768) if ti is None else (2, 1, 768))
            torch.testing.assert_close(out, g["outs"][key], rtol=1e-5, atol=1e-5)
