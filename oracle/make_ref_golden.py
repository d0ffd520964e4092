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
import sys

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


def infer_golden():
    """``run_inference`` (models/infer.py:7-123) EXECUTED over the oracle's tiny models (``oracle/infer_case.py``): the processors are the REFERENCE
    class installed by the REFERENCE ``set_visual_cross_attention_adapter``, the adapters the real ``models.adapters.PhotoVerseAdapter``, the
    sampler ``ref_exec.SchedulerStandIn``.  Stored per case: the returned images, the tensor handed to ``vae.decode`` (= final latents /
    scaling_factor, captured by a recording wrapper), the scheduler call log, the tokenizer calls, and for training_mode the fusion draws."""
    import sys

    from oracle import infer_case as ic
    run_inference = ref_exec.reference_run_inference()
    set_adapter, _ = ref_exec.reference_unet_helpers()
    sys.path.insert(0, ref_exec.REF_ROOT)
    from models.adapters import PhotoVerseAdapter
    sys.path.pop(0)
    m = ic.oracle_models(processor_installer=set_adapter)
    m["image_adapter"] = PhotoVerseAdapter(clip_embedding_dim=ic.VIS["hidden_size"], cross_attention_dim=768, num_tokens=ic.NUM_TOKENS).eval()
    m["text_adapter"] = PhotoVerseAdapter(clip_embedding_dim=ic.VIS["hidden_size"], cross_attention_dim=768, num_tokens=ic.NUM_TOKENS).eval()
    ic.fill_all_(image_adapter=m["image_adapter"], text_adapter=m["text_adapter"])
    for mod in m.values():
        mod.requires_grad_(False)                                    # modeling_utils.py:63-66 (adapters: inference)

    class RecordingVAE:
        """delegates to the oracle VAE; remembers what ``infer.py:122`` decodes and what ``:63`` encodes"""
        def __init__(self, vae):
            self._vae, self.config, self.decoded, self.encoded = vae, vae.config, None, None

        def encode(self, x):
            self.encoded = x.clone()
            return self._vae.encode(x)

        def decode(self, z):
            self.decoded = z.detach().clone()
            return self._vae.decode(z)

    from types import SimpleNamespace
    ddpm = SimpleNamespace(config=dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                       steps_offset=1, timestep_spacing="leading", prediction_type="epsilon"))
    n_attn2 = sum(1 for n in m["unet"].attn_processors if n.endswith("attn2.processor"))
    out = {"seeds": dict(ic.SEEDS), "checksums": {k: checksums(v) for k, v in m.items()}, "n_attn2": n_attn2, "cases": {}}
    for name, case in ic.CASES.items():
        kw = case["kw"]
        tok, vae = ic.TokenizerStub(), RecordingVAE(m["vae"])
        ex = ic.example(case["negative"])
        ref_exec.SchedulerStandIn.log = []
        if "global_seed" in case:
            torch.manual_seed(case["global_seed"])
        with torch.no_grad():                                         # generate.py:80 calls it under no_grad
            images = run_inference(ex, tok, m["image_encoder"], m["text_encoder"], m["unet"], m["text_adapter"], m["image_adapter"], vae,
                                   ddpm, "cpu", ic.LAYERS_IDX, **kw)
        rec = {"images": images.detach().clone(), "decode_input": vae.decoded, "scheduler_log": list(ref_exec.SchedulerStandIn.log),
               "tokenizer_calls": list(tok.calls), "encoded_pixel_values": vae.encoded is not None}
        if kw.get("training_mode"):
            # the processors' grad-mode draws (attention_processor.py:414) come from the global generator the noise draw seeded (:57):
            # replay the stream to store them - uncond forward's attn2 layers in call order, then the cond forward's
            g = torch.manual_seed(kw["seed"])
            torch.randn((ic.BATCH, 4, ic.LATENT, ic.LATENT), generator=g)
            rec["fusion_draws"] = [torch.rand(1).item() for _ in range(2 * n_attn2)]
        out["cases"][name] = rec
        print(f"  infer case {name}: images {tuple(images.shape)}, |decode_input| {vae.decoded.norm():.4f}, sched calls {len(rec['scheduler_log'])}")
    torch.save(out, os.path.join(OUT, "ref_infer_golden.pt"))


CKPT_SEEDS = dict(unet=171, image_adapter=172, text_adapter=173, other=181)
CKPT_LORA = dict(r=4, lora_alpha=8, lora_dropout=0.0, bias="none", target_modules=["attn2.to_k", "attn2.to_v", "attn2.to_q"])    # train.py:348-354


def _sd_inventory(sd):
    return {k: (tuple(v.shape), v.double().sum().item(), (v.double() ** 2).sum().item()) for k, v in sd.items()}


def checkpoint_golden():
    """``save_progress`` / ``load_photoverse_model`` (models/modeling_utils.py:13-50) EXECUTED on the oracle's tiny UNet (reference processors
    installed by the reference function; LoRA through the peft stand-in) and the real adapters.  The files themselves are ~100 MB, so the fixture
    stores their INVENTORY - file names, top-level key order, per-section key lists with shape + checksums (weights are seeded by name,
    ``oracle.seeded.fill_state_``, so a test regenerates them) - plus the eps of the UNet ``load_photoverse_model`` returned."""
    import sys
    import tempfile

    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef
    load_pv, save_progress = ref_exec.reference_checkpoint_functions()
    set_adapter, _ = ref_exec.reference_unet_helpers()
    sys.path.insert(0, ref_exec.REF_ROOT)
    from models.adapters import PhotoVerseAdapter
    sys.path.pop(0)

    def build(lora, seeds):
        torch.manual_seed(0)
        unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
        unet.requires_grad_(False)
        set_adapter(unet, num_tokens=(2,))
        if lora:
            ref_exec._inject_stand_in(ref_exec.LoraConfigStandIn(**CKPT_LORA), unet)
        ia = PhotoVerseAdapter(clip_embedding_dim=64, cross_attention_dim=768, num_tokens=2)
        ta = PhotoVerseAdapter(clip_embedding_dim=64, cross_attention_dim=768, num_tokens=2)
        fill_state_(unet, seeds["unet"]); fill_state_(ia, seeds["image_adapter"]); fill_state_(ta, seeds["text_adapter"])
        return unet, ia, ta

    class DDPLike(torch.nn.Module):                      # what accelerator.prepare hands back under DDP; unwrap_model strips it
        def __init__(self, module):
            super().__init__()
            self.module = module

    out = {"seeds": CKPT_SEEDS, "lora": CKPT_LORA, "files": {}}
    g = torch.Generator().manual_seed(190)
    x, text, ip = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 77, 768, generator=g), torch.randn(2, 2, 768, generator=g)
    out["probe"] = {"x": x, "text": text, "ip": ip, "t": 500}
    with tempfile.TemporaryDirectory() as tmp:
        # (1) no LoRA, final save (step=None), no optimizer: train.py:627 form
        unet, ia, ta = build(False, CKPT_SEEDS)
        save_progress(ia, ta, unet, ref_exec.AcceleratorStandIn, tmp)
        # (2) LoRA, periodic save (step=7) with optimizer, models wrapped like accelerator.prepare's DDP: train.py:608 form
        unet_l, ia_l, ta_l = build(True, CKPT_SEEDS)
        trainable = [p for n, p in unet_l.named_parameters() if p.requires_grad] + list(ia_l.parameters()) + list(ta_l.parameters())
        opt = torch.optim.AdamW(trainable, lr=1e-4)
        lcfg = ref_exec.LoraConfigStandIn(**CKPT_LORA)
        save_progress(DDPLike(ia_l), DDPLike(ta_l), DDPLike(unet_l), ref_exec.AcceleratorStandIn, tmp, step=7, lora_config=lcfg, optimizer=opt)
        out["listdir"] = sorted(os.listdir(tmp))
        out["trainable_unet_names_lora"] = [n for n, p in unet_l.named_parameters() if p.requires_grad]
        for fname in out["listdir"]:
            sd = torch.load(os.path.join(tmp, fname), map_location="cpu", weights_only=False)
            inv = {"top_level_keys": list(sd.keys())}
            for sec in ("image_adapter", "text_adapter", "cross_attention_adapter"):
                inv[sec] = _sd_inventory(sd[sec])
            if "lora_config" in sd:
                inv["lora_config"] = sd["lora_config"]
            if "optimizer" in sd:
                inv["optimizer_keys"] = sorted(sd["optimizer"].keys())
            inv["n_unet_keys_total"] = len((unet_l if "lora_config" in sd else unet).state_dict())
            out["files"][fname] = inv
        # (3) load_photoverse_model round trip: FRESH models with other numbers and NO LoRA; the loader injects it from the file's lora_config
        unet2, ia2, ta2 = build(False, {k: CKPT_SEEDS["other"] for k in CKPT_SEEDS})
        ia3, ta3, unet3, lora_cfg = load_pv(os.path.join(tmp, "photoverse_000007.pt"), ia2, ta2, unet2)
        assert ia3 is ia2 and ta3 is ta2 and unet3 is unet2
        for k, v in ia_l.state_dict().items():
            assert torch.equal(ia3.state_dict()[k], v)
        saved = torch.load(os.path.join(tmp, "photoverse_000007.pt"), weights_only=False)["cross_attention_adapter"]
        for k, v in saved.items():
            assert torch.equal(unet3.state_dict()[k], v), k
        with torch.no_grad():
            eps_loaded = unet3(x, torch.tensor(500), encoder_hidden_states=(text, ip)).sample
            # the same probe on the model the file was saved FROM differs: the file only carries the attn2 subset (:34-37)
            eps_source = unet_l(x, torch.tensor(500), encoder_hidden_states=(text, ip)).sample
        out["load"] = {"returned_lora_config": lora_cfg.to_dict(), "eps_loaded": eps_loaded, "eps_source": eps_source,
                       "lora_modules": sorted(n for n, m_ in unet3.named_modules() if hasattr(m_, "lora_A")),
                       "unet_state_keys_after_load": list(unet3.state_dict().keys())}
        # no lora_config in the file -> returned config is None, UNet untouched structurally
        unet4, ia4, ta4 = build(False, {k: CKPT_SEEDS["other"] for k in CKPT_SEEDS})
        _, _, unet5, none_cfg = load_pv(os.path.join(tmp, "photoverse.pt"), ia4, ta4, unet4)
        out["load_nolora"] = {"returned_lora_config": none_cfg, "lora_modules": sorted(n for n, m_ in unet5.named_modules() if hasattr(m_, "lora_A"))}
    torch.save(out, os.path.join(OUT, "ref_checkpoint_golden.pt"))
    print("  files:", out["listdir"], {f: v["top_level_keys"] for f, v in out["files"].items()})
    print("  cross_attention_adapter keys:", {f: len(v["cross_attention_adapter"]) for f, v in out["files"].items()}, "lora modules", len(out["load"]["lora_modules"]))


def load_models_golden():
    """``load_models`` (models/modeling_utils.py:53-95) EXECUTED with the six ``from_pretrained`` classes bound to recording factories that
    return the oracle's tiny models: the order and arguments of the downloads, what is frozen, the adapters' sizes, the processors installed,
    the LoRA assertion, the 9-tuple order, and the photoverse_path branch."""
    import tempfile
    from types import SimpleNamespace

    from oracle import infer_case as ic
    from oracle.clip_ref import CLIPTextModelRef, CLIPVisionModelRef
    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef
    from oracle.vae_ref import AutoencoderKLDecoderRef
    factories = {"CLIPTokenizer": ic.TokenizerStub, "CLIPTextModel": lambda: CLIPTextModelRef(**ic.TXT),
                 "AutoencoderKL": lambda: AutoencoderKLDecoderRef(**ic.VAE), "UNet2DConditionModel": lambda: UNet2DConditionModelRef(**TINY_CONFIG),
                 "CLIPVisionModel": lambda: CLIPVisionModelRef(**ic.VIS), "DDPMScheduler": lambda: SimpleNamespace(config={"num_train_timesteps": 1000})}
    out = {"runs": {}}

    def describe(ret, log):
        tokenizer, text_encoder, vae, unet, image_encoder, image_adapter, text_adapter, scheduler, lora_config = ret
        return {"call_log": list(log), "tuple_types": [type(v).__name__ for v in ret],
                "unet_trainable": [n for n, p in unet.named_parameters() if p.requires_grad],
                "unet_n_params": len(list(unet.named_parameters())),
                "frozen": {k: not any(p.requires_grad for p in m.parameters()) for k, m in (("vae", vae), ("text_encoder", text_encoder), ("image_encoder", image_encoder))},
                "adapters_trainable": all(p.requires_grad for a in (image_adapter, text_adapter) for p in a.parameters()),
                "image_adapter_keys": list(image_adapter.state_dict().keys()), "text_adapter_keys": list(text_adapter.state_dict().keys()),
                "adapter_in_features": image_adapter.mapping_0[0].in_features, "adapter_out_features": image_adapter.mapping_0[6].out_features,
                "processors": {n: (type(p).__name__, tuple(getattr(p, "num_tokens", ()) or ())) for n, p in unet.attn_processors.items()},
                "lora_config": None if lora_config is None else lora_config.to_dict()}

    log = []
    load_models = ref_exec.reference_load_models(factories, log)
    out["runs"]["extra1"] = describe(load_models("some/model-id", 1), log)
    log.clear()
    cfg = ref_exec.LoraConfigStandIn(**CKPT_LORA)
    out["runs"]["extra4_lora"] = describe(load_models("some/model-id", 4, use_lora=True, lora_config=cfg), log)
    try:
        load_models("some/model-id", 1, use_lora=True)
        out["lora_assert"] = None
    except AssertionError as e:
        out["lora_assert"] = str(e)
    # photoverse_path: a checkpoint with a lora_config overrides the (absent) argument (:90-92)
    _, save_progress = ref_exec.reference_checkpoint_functions()
    with tempfile.TemporaryDirectory() as tmp:
        log.clear()
        ret = load_models("some/model-id", 1, use_lora=True, lora_config=cfg)
        save_progress(ret[5], ret[6], ret[3], ref_exec.AcceleratorStandIn, tmp, lora_config=cfg)
        log.clear()
        out["runs"]["extra1_from_checkpoint"] = describe(load_models("some/model-id", 1, photoverse_path=os.path.join(tmp, "photoverse.pt")), log)
    torch.save(out, os.path.join(OUT, "ref_load_models_golden.pt"))
    for k, v in out["runs"].items():
        print(f"  load_models[{k}]: calls {[c[0] for c in v['call_log']]}, trainable unet params {len(v['unet_trainable'])}/{v['unet_n_params']}, lora {v['lora_config'] is not None}")
    print("  lora assert:", out["lora_assert"])


def processor640_golden():
    """The reference processor at the C = 640 / d = 80 level (SD-v1.5's 32x32 blocks) with N = 128 rows per sample, so the HIP side of the
    comparison is the ONE-LAUNCH fused attn2 kernel's C = 640 instantiation (N % 128 == 0), not the four-launch path: no_grad sum, P in {1, 5},
    and the two grad-mode 2x branches as forward values.  Inputs fp16-representable; expected outputs stored as fp32."""
    from oracle.unet_ref import AttentionRef
    _, Proc = ref_exec.reference_attention_processors()
    C, heads, N, B = 640, 8, 128, 2
    torch.manual_seed(17)
    attn = AttentionRef(C, cross_attention_dim=768, heads=heads, dim_head=C // heads).eval()
    fill_state_(attn, 241)
    out = {"C": C, "heads": heads, "N": N, "attn_seed": 241, "proc_seed": 243, "attn_checksums": checksums(attn), "cases": {}}
    for P in (1, 5):
        proc = Proc(hidden_size=C, cross_attention_dim=768, num_tokens=(P,))
        fill_state_(proc, 243)
        g = torch.Generator().manual_seed(300 + P)
        hs16, text16, ip16 = (torch.randn(*shp, generator=g).half() for shp in ((B, N, C), (B, 77, 768), (B, P, 768)))
        hs16[:, :, ::7] += 1.5
        gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
        beta = 0.1 * torch.randn(C, generator=g)
        hs, text, ip = hs16.float(), text16.float(), ip16.float()
        normed = torch.nn.functional.layer_norm(hs, (C,), gamma, beta, 1e-5)       # [EXT] BasicTransformerBlock.norm2 in front of attn2
        case = {"hs": hs16, "text": text16, "ip": ip16, "gamma": gamma, "beta": beta, "subsample": "outputs[:, ::2] (every second query row)"}
        with torch.no_grad():
            case["nograd"] = proc(attn, hs, encoder_hidden_states=(text, ip))[:, ::2].clone()
            case["vnorm"] = proc.to_v_ip_norm.clone()
            case["nograd_on_normed"] = proc(attn, normed, encoder_hidden_states=(text, ip))[:, ::2].clone()
        for region in (("text", "ip") if P == 5 else ()):
            seed, u = _seed_for(region)
            torch.manual_seed(seed)
            with torch.enable_grad():
                case["grad_" + region] = {"u": u, "out": proc(attn, hs, encoder_hidden_states=(text, ip)).detach()[:, ::2].clone()}
        out["cases"][P] = case
    torch.save(out, os.path.join(OUT, "ref_processor640_golden.pt"))


def unet32_golden():
    """As ``unet_golden`` but on a 32x32 latent: the tiny UNet's 640-wide mid-block attention then has 16 x 16 = 256 rows, which routes the HIP
    model through the C = 640 fused attn2 kernel (and the 320-wide levels through the C = 320 one at 1024 rows)."""
    from oracle.unet_ref import TINY_CONFIG, UNet2DConditionModelRef
    set_adapter, get_vnorm = ref_exec.reference_unet_helpers()
    torch.manual_seed(0)
    unet = UNet2DConditionModelRef(**TINY_CONFIG).eval()
    set_adapter(unet, num_tokens=(1,))
    fill_state_(unet, 257)
    g = torch.Generator().manual_seed(258)
    x = torch.randn(2, 4, 32, 32, generator=g)
    text, ip = torch.randn(2, 77, 768, generator=g).half().float(), torch.randn(2, 1, 768, generator=g)
    with torch.no_grad():
        eps = unet(x, torch.tensor(641), encoder_hidden_states=(text, ip)).sample
        vnorm = get_vnorm(unet)
    torch.save({"weights_seed": 257, "checksums": checksums(unet), "x": x, "text": text.half(), "ip": ip, "t": 641, "eps": eps, "vnorm": vnorm},
               os.path.join(OUT, "ref_unet32_golden.pt"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for fn in [globals()[a] for a in sys.argv[1:]] or (inject_golden, text_golden, processor_golden, unet_golden, arcface_golden, adapter17_golden,
               infer_golden, checkpoint_golden, load_models_golden, processor640_golden, unet32_golden):
        fn()
        print("wrote", fn.__name__)
    print("stand-ins used:")
    for k, v in ref_exec.STAND_INS.items():
        print(" ", k, "-", v)
