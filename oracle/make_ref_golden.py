"""Generates ``tests/golden/ref_*.pt`` by EXECUTING THE REFERENCE'S OWN CODE (``oracle/ref_exec.py``: definitions loaded from the
source text under ``/root/reference``, nothing rewritten).  Run in the BUILD container only: ``python -m oracle.make_ref_golden``.

Fixtures are data only - seeds, small inputs, expected outputs; weights are never stored when a seed reproduces them
(``oracle/seeded.fill_state_`` is order-independent, so the real class here and the restatement / HIP model in the tests hold the same
numbers; per-tensor checksums are stored so a test can prove it).

  ref_inject_golden.pt     ``_inject_concept_embeddings`` (models/clip.py:17-24): E in {1, 5}, idx as (B,1) and (B,), edge rows
  ref_text_golden.pt       ``clip_text_transformer_forward`` (models/clip.py:29-102) over the installed transformers CLIPTextModel's own
                           submodules (tiny config; weights stored: 0.5 MB) with and without injected concept embeddings
  ref_processor_golden.pt  ``PhotoVerseAttnProcessor2_0.__call__`` (models/attention_processor.py:245-435) with the oracle's
                           ``AttentionRef`` as ``attn``: no_grad sum, the three grad-mode fusion branches (``torch.rand(1).item()`` under
                           chosen global seeds), tuple / list / deprecated bare-tensor conventions, P in {1, 5}, gradients
  ref_unet_golden.pt       ``set_visual_cross_attention_adapter`` + ``get_visual_cross_attention_values_norm`` (models/unet.py:8-47) on the
                           oracle's tiny UNet with the REFERENCE processor class installed: eps, V-norm stack, processor inventory
  ref_adapter17_golden.pt  the real ``PhotoVerseAdapter`` with 17 mapping pairs on 6 hidden states (BASELINE configs[4] conditioning shape)
  ref_arcface_golden.pt    ``ArcFaceResNet18(pretrained=False)`` (models/arcface_resnet.py:12-134) and ``FaceLoss.preprocess`` / ``forward``
                           (models/loss.py:26-78): embeddings, preprocess output, loss, d loss / d x_gen
"""
import os

import torch

from oracle import ref_exec
from oracle.seeded import checksums, fill_state_

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def inject_golden():
    inject, _ = ref_exec.reference_clip_functions()
    g = torch.Generator().manual_seed(21)
    cases = []
    for E, idx in ((5, torch.tensor([[5], [1], [71]])), (1, torch.tensor([[5], [1], [76]])), (5, torch.tensor([9, 72, 0])),
                   (3, torch.tensor([[74], [0], [33]]))):
        old = torch.randn(3, 77, 16, generator=g)
        concept = torch.randn(3, E, 16, generator=g)
        cases.append({"old": old, "concept": concept, "idx": idx, "expected": inject(old, concept, idx)})
    torch.save({"cases": cases}, os.path.join(OUT, "ref_inject_golden.pt"))


def text_golden():
    from transformers import CLIPTextConfig, CLIPTextModel
    _, fwd = ref_exec.reference_clip_functions()
    cfg = CLIPTextConfig(vocab_size=120, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                         max_position_embeddings=77, hidden_act="quick_gelu", bos_token_id=118, eos_token_id=119, pad_token_id=0)
    torch.manual_seed(5)
    hf = CLIPTextModel(cfg).eval()
    fill_state_(hf, 31)
    shim = ref_exec.TextTransformerShim(hf)
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(1, 118, (3, 77), generator=g)
    ids[:, 0] = 118
    ids[0, 20:] = 119
    ids[1, 9:] = 119
    ids[2, 76] = 119
    outs = {}
    with torch.no_grad():
        for E, idx in ((0, None), (1, torch.tensor([[5], [1], [40]])), (5, torch.tensor([[5], [1], [71]]))):
            d = {"text_input_ids": ids}
            if E:
                concept = torch.randn(3, E, 128, generator=g)
                d.update(concept_text_embeddings=concept, concept_placeholder_idx=idx)
            o = fwd(shim, d)
            outs[E] = {"concept": d.get("concept_text_embeddings"), "idx": idx, "last_hidden_state": o[0].clone(), "pooled": o[1].clone()}
        try:
            fwd(shim, None)
            raised = None
        except ValueError as e:
            raised = str(e)
    torch.save({"config": dict(vocab_size=120, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                               max_position_embeddings=77),
                "state_dict": {k: v.clone() for k, v in hf.state_dict().items()}, "ids": ids, "outs": outs, "none_error": raised},
               os.path.join(OUT, "ref_text_golden.pt"))


def _seed_for(region):
    """A global torch seed whose first ``torch.rand(1).item()`` lands in the requested fusion region (attention_processor.py:414-420)."""
    for s in range(1000):
        torch.manual_seed(s)
        u = torch.rand(1).item()
        if (region == "text" and u < 1 / 3 - 0.05) or (region == "ip" and u > 2 / 3 + 0.05) or (region == "sum" and 0.4 < u < 0.6):
            return s, u
    raise RuntimeError(region)


def processor_golden():
    from oracle.unet_ref import AttentionRef
    _, Proc = ref_exec.reference_attention_processors()
    C, heads, N, B = 320, 8, 64, 2
    torch.manual_seed(17)
    attn = AttentionRef(C, cross_attention_dim=768, heads=heads, dim_head=C // heads).eval()
    fill_state_(attn, 41)
    out = {"C": C, "heads": heads, "attn_seed": 41, "proc_seed": 43, "attn_checksums": checksums(attn), "cases": {}}
    for P in (1, 5):
        proc = Proc(hidden_size=C, cross_attention_dim=768, num_tokens=(P,))
        fill_state_(proc, 43)
        if P == 1:
            out["proc_checksums"] = checksums(proc)
        g = torch.Generator().manual_seed(100 + P)
        # inputs are fp16-representable and stored as fp16 (file size); big gradients are stored as strided sub-samples
        hs16, text16, ip16, G16 = (torch.randn(*shp, generator=g).half() for shp in ((B, N, C), (B, 77, 768), (B, P, 768), (B, N, C)))
        hs, text, ip, G = hs16.float(), text16.float(), ip16.float(), G16.float()
        case = {"hs": hs16, "text": text16, "ip": ip16, "G": G16, "subsample": "d_text[:, :, ::8]; d_to_k_ip / d_to_v_ip [::4, ::4]"}
        with torch.no_grad():
            case["nograd_tuple"] = proc(attn, hs, encoder_hidden_states=(text, ip))
            case["vnorm"] = proc.to_v_ip_norm.clone()
            case["nograd_list"] = proc(attn, hs, encoder_hidden_states=(text, [ip]))
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                case["nograd_tensor"] = proc(attn, hs, encoder_hidden_states=torch.cat([text, ip], dim=1))
        for region in ("text", "sum", "ip"):
            seed, u = _seed_for(region)
            ps = [proc.to_k_ip[0].weight, proc.to_v_ip[0].weight]
            for p_ in ps:
                p_.requires_grad_(True)
                p_.grad = None
            h, t, i = (v.clone().requires_grad_(True) for v in (hs, text, ip))
            torch.manual_seed(seed)
            with torch.enable_grad():
                o = proc(attn, h, encoder_hidden_states=(t, i))
                loss = (o * G).sum() + 0.3 * proc.to_v_ip_norm.sum()
            loss.backward()
            def gr(v):                              # a branch that does not reach the loss leaves .grad = None (e.g. to_k_ip when u < 1/3)
                return torch.zeros_like(v) if v.grad is None else v.grad.clone()
            case["grad_" + region] = {"u": u, "torch_seed": seed, "out": o.detach().clone(), "d_hs": gr(h), "d_text": gr(t)[:, :, ::8].clone(),
                                      "d_ip": gr(i), "d_to_k_ip": gr(ps[0])[::4, ::4].clone(), "d_to_v_ip": gr(ps[1])[::4, ::4].clone()}
        out["cases"][P] = case
    # validation errors of __init__ (attention_processor.py:37-49)
    errs = {}
    for key, kw in (("fusion_type", dict(fusion_rules=[1 / 3, 2 / 3])), ("fusion_sum", dict(fusion_rules=(0.5, 0.6))),
                    ("scale_len", dict(scale=[1.0, 2.0]))):
        try:
            Proc(hidden_size=C, cross_attention_dim=768, num_tokens=(5,), **kw)
            errs[key] = None
        except ValueError as e:
            errs[key] = str(e)
    out["init_errors"] = errs
    torch.save(out, os.path.join(OUT, "ref_processor_golden.pt"))


def unet_golden():
    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef
    set_adapter, get_vnorm = ref_exec.reference_unet_helpers()
    torch.manual_seed(0)
    unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_adapter(unet, num_tokens=(5,))          # the REFERENCE function installs the REFERENCE processor class
    fill_state_(unet, 57)
    g = torch.Generator().manual_seed(58)
    x = torch.randn(2, 4, 16, 16, generator=g)
    text, ip = torch.randn(2, 77, 768, generator=g), torch.randn(2, 5, 768, generator=g)
    with torch.no_grad():
        eps = unet(x, torch.tensor(321), encoder_hidden_states=(text, ip)).sample
        vnorm = get_vnorm(unet)
    inv = {n: (type(p).__name__, getattr(p, "hidden_size", None), getattr(p, "cross_attention_dim", None)) for n, p in unet.attn_processors.items()}
    torch.save({"weights_seed": 57, "checksums": checksums(unet), "x": x, "text": text, "ip": ip, "t": 321, "eps": eps, "vnorm": vnorm,
                "processors": inv}, os.path.join(OUT, "ref_unet_golden.pt"))


def arcface_golden():
    Net, FaceLoss = ref_exec.reference_arcface()
    torch.manual_seed(3)
    fl = FaceLoss("cpu", model_name="arcface")
    fill_state_(fl.model, 71)
    assert not fl.model.training
    g = torch.Generator().manual_seed(72)
    gray = torch.randn(2, 1, 128, 128, generator=g)
    x = torch.rand(2, 3, 96, 80, generator=g) * 255.0
    x_gen = (x + 40.0 * torch.randn(2, 3, 96, 80, generator=g)).clamp(0, 255)
    with torch.no_grad():
        emb = fl.model(gray)
        pre = fl.preprocess(x)
        pre_raw = fl.preprocess(x[:, :1], normalize=False)
        loss_min = fl(x, x_gen, maximize=False)
    xg = x_gen.clone().requires_grad_(True)
    loss = fl(x, xg)
    loss.backward()
    torch.save({"weights_seed": 71, "checksums": checksums(fl.model), "state_keys": list(fl.model.state_dict().keys()), "gray": gray,
                "embedding": emb, "x": x, "x_gen": x_gen, "preprocess": pre, "preprocess_raw_1ch": pre_raw, "loss": loss.detach(),
                "loss_minimize": loss_min, "d_x_gen": xg.grad.clone()}, os.path.join(OUT, "ref_arcface_golden.pt"))


def adapter17_golden():
    """BASELINE configs[4] conditioning shape: extra_num_tokens = 16 -> 17 mapping pairs, encoder_layers_idx = 4,8,12,16,20 -> 6 CLIP hidden states
    (infer.py:80-84), executed on the REAL reference class (models/adapters.py imports without diffusers).  Inputs are re-drawn from the seed by the tests."""
    import sys
    sys.path.insert(0, ref_exec.REF_ROOT)
    from models.adapters import PhotoVerseAdapter
    sys.path.pop(0)
    ad = PhotoVerseAdapter(clip_embedding_dim=1024, cross_attention_dim=768, num_tokens=17).eval()
    fill_state_(ad, 91)
    g = torch.Generator().manual_seed(92)
    embs = [torch.randn(2, 257, 1024, generator=g).half().float() for _ in range(6)]
    with torch.no_grad():
        outs = {"none": ad(embs), "0": ad(embs, token_index=0), "5": ad(embs, token_index=5)}
    torch.save({"weights_seed": 91, "input_seed": 92, "n_state": len(ad.state_dict()), "outs": outs}, os.path.join(OUT, "ref_adapter17_golden.pt"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for fn in (inject_golden, text_golden, processor_golden, unet_golden, arcface_golden, adapter17_golden):
        fn()
        print("wrote", fn.__name__)
    print("stand-ins used:")
    for k, v in ref_exec.STAND_INS.items():
        print(" ", k, "-", v)
