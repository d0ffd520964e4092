"""CPU-only tests: the C-ABI library loads and exports every declared symbol, host logic (scheduler tables, weight
packing, tokenizer, checkpoint layout, LoRA naming, loaders), the product path refuses to run without a HIP device,
and the N>1 sharding / gather path under gloo (world_size 2)."""
import os
import re
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = dict(in_channels=4, out_channels=4, block_out_channels=(320, 640), layers_per_block=1,
            down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"), up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"),
            attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5)
VIS = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=1, image_size=56, patch_size=14)
VAE = dict(block_out_channels=(128, 256), layers_per_block=1)
TXT = dict(vocab_size=1000, hidden_size=768, num_attention_heads=12, intermediate_size=256, num_hidden_layers=1)


@pytest.fixture(scope="module")
def lib():
    from photoverse_amd.build import build_lib
    build_lib(verbose=False)              # hipcc cross-compiles gfx950 without a GPU
    from photoverse_amd import _lib
    return _lib.load()


def test_cabi_exports_every_declared_symbol(lib):
    from photoverse_amd import _lib
    header = open(os.path.join(ROOT, "include", "photoverse_hip.h")).read()
    declared = set(re.findall(r"^int\s+(pv_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.pv_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define PV_ABI_VERSION (\d+)", header).group(1))
    assert lib.pv_device_count() in (-1, 0) or torch.cuda.is_available()


def test_cabi_rejects_bad_parameter_blocks_before_touching_the_device(lib):
    """Error behaviour of the C-ABI (no GPU needed: every entry point validates its parameter block before its first HIP call and returns
    hipErrorInvalidValue = 1): empty blocks, and the shape limits the header documents for the attn2 entry points and pv_row_gemm."""
    import ctypes as C
    from photoverse_amd import _lib
    INVALID = 1
    for name, cls in (("pv_gemm_conv", _lib.GemmParams), ("pv_attention", _lib.AttnParams), ("pv_cross_attention", _lib.XAttnParams),
                      ("pv_cross_attention_fused", _lib.XAttnFusedParams), ("pv_cross_attention_lnq", _lib.XAttnLnqParams), ("pv_row_gemm", _lib.RowGemmParams),
                      ("pv_groupnorm_apply", _lib.GroupNormParams), ("pv_layernorm", _lib.LayerNormParams), ("pv_layernorm_backward", _lib.LayerNormBwdParams),
                      ("pv_attention_backward", _lib.AttnBwdParams), ("pv_cross_attention_backward", _lib.XAttnBwdParams),
                      ("pv_groupnorm_backward", _lib.GroupNormBwdParams)):
        assert getattr(lib, name)(C.byref(cls()), None) == INVALID, name
    FAKE = 0x1000                                    # never dereferenced: the shape checks come first

    def call(fn, cls, ptrs, base, **kw):
        p = cls()
        for k in ptrs:
            setattr(p, k, FAKE)
        for k, v in {**base, **kw}.items():
            setattr(p, k, v)
        return fn(C.byref(p), None)

    lnq = lambda **kw: call(lib.pv_cross_attention_lnq, _lib.XAttnLnqParams, ("hs", "wq", "q_bias", "wq_rowsum", "kt", "vt", "kip", "vip", "out"),
                            dict(ld_hs=1280, ln=1, ln_eps=1e-5, ldkt=2560, ldvt=2560, ldkip=2560, ldvip=2560, ldo=1280, batch=2, nq=256, heads=8, d=160, nt=77, nip=1,
                                 w_text=1.0, w_ip=1.0), **kw)
    for bad in (dict(d=40), dict(nt=81), dict(nip=17), dict(ld_hs=1283), dict(wq_rowsum=0), dict(batch=0)):      # d in {160, 80}; nt <= 80; nip <= 16; 16-byte rows; ln needs the row sums
        assert lnq(**bad) == INVALID, bad
    # 32-bit buffer offsets: an activation extent of 2 GiB or more is rejected, not read as zeros
    assert lnq(batch=4096, nq=1024, ld_hs=1280) == INVALID and lnq(batch=4096, nq=1024, ldo=1280) == INVALID
    lnb = lambda **kw: call(lib.pv_layernorm_backward, _lib.LayerNormBwdParams, ("x", "dy", "dx", "gamma", "beta", "add"),
                            dict(ldx=320, lddy=320, lddx=320, ldadd=320, rows=64, cols=320, eps=1e-5), **kw)
    assert lnb(ldadd=300) == INVALID                 # the accumulated gradient's rows must hold a whole row
    fused = lambda **kw: call(lib.pv_cross_attention_fused, _lib.XAttnFusedParams, ("hs", "wq", "wo", "kimg", "vimg", "out"),
                              dict(ld_hs=320, ld_out=320, batch=2, nq=256, heads=8, d=40, nt=77, nip=1, w_text=1.0, w_ip=1.0), **kw)
    for bad in (dict(d=48), dict(nq=200), dict(heads=4), dict(nip=0), dict(nt=60), dict(nip=17)):                # C in {320, 640}; nq % 128; 8 heads; 64 < nt <= 80; 1 <= nip <= 16
        assert fused(**bad) == INVALID, bad
    abw = lambda **kw: call(lib.pv_attention_backward, _lib.AttnBwdParams, ("q", "k", "v", "out", "dout", "lse", "delta", "qs", "dq", "dk", "dv"),
                            dict(ldq=960, ldk=960, ldv=960, ldo=320, lddo=320, ldqs=320, lddq=960, lddk=960, lddv=960, batch=2, heads=8, nq=512, nk=512, d=40), **kw)
    for bad in (dict(ldq=324), dict(nq=0), dict(d=48), dict(ws=FAKE, ws_bytes=-1)):       # 16-byte rows; d in {40, 64, 80, 160}; a workspace has a size
        assert abw(**bad) == INVALID, bad
    rowg = lambda **kw: call(lib.pv_row_gemm, _lib.RowGemmParams, ("x", "w", "out"), dict(ld_x=320, M=4096, K=320, N=960, ln=1, ln_eps=1e-5, geglu=0, ld_out=960), **kw)
    for bad in (dict(K=640, ld_x=640), dict(N=1000, ld_out=1000), dict(ld_out=320), dict(M=0)):                   # K == 320; N % 320 == 0; ld_out >= N
        assert rowg(**bad) == INVALID, bad
    # the GroupNorm fold (pv_gemm_params.a_norm) exists for 3x3 convs only, with NONE or SILU behind the affine part: a Linear launch carrying it, or
    # another activation code, is rejected up front instead of running with the fold silently ignored
    gemm = lambda **kw: call(lib.pv_gemm_conv, _lib.GemmParams, ("a0", "w", "out", "a_norm"),
                             dict(lda0=640, c0=640, batch=16, hin=64, win=64, hout=64, wout=64, taps=1, stride=1, pad=0, N=320, ldc=320, M=65536,
                                  a_norm_act=1), **kw)                  # PV_ACT_SILU
    assert gemm() == INVALID                                                                              # Linear (taps == 1) + a_norm
    assert gemm(taps=9, pad=1, lda0=320, c0=320, a_norm_act=2) == INVALID    # 3x3 conv + PV_ACT_QUICK_GELU: an activation the fold does not have


def test_no_mixed_shape_mfma_chain_in_the_shipped_isa():
    """A 16x16x16 tail MFMA that hipcc put ONE instruction behind the 16x16x32 whose result it accumulates onto returned wrong sums, differently per
    run (round 5, EXPERIMENTS.md; root cause not established).  Every kernel that chains the two shapes fences its groups; this scans the ISA hipcc
    emits today for the two sources that hold such chains (every instantiated variant, the default attn8_kernel<497> and the backward passes included)
    and requires that no dependent pair of different shapes sits within three instructions of each other.  hipcc cross-compiles: CPU only, ~15 s."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mfma_chain_scan", os.path.join(ROOT, "tools", "diag", "mfma_chain_scan.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    hits, n_mfma = scan.scan(scan.TWO_SHAPE_SOURCES, dist=3)
    assert n_mfma > 1000, "the scan saw no MFMA instructions: it scanned nothing"
    assert not hits, "\n".join("%s %s: %s -> %s, %d between (%s) x%d" % (*k[:5], ", ".join(k[5]), v) for k, v in hits.items())


def test_kernel_info_names_the_launch_without_a_gpu(lib):
    """ABI 17: ``pv_gemm_conv_kernel_info`` / ``pv_attention_kernel_info`` answer from the launchers' own validation + dispatch code in describe-only mode - no
    HIP call, so the headline's dispatch decisions are pinned here on the CPU: 64 x 64 convs on the LDS-resident-patch instantiation of the one-per-CU
    tile, half-chip launches only with the side-by-side threshold, short-K Linear layers and the 8 x 8 convs (split-K) on the 128-row kernel, GEGLU on the
    256-column tile, d = 40 self-attention on the 8-wave kernel from one workgroup per CU on; and a rejected block is rejected by the query too."""
    from photoverse_amd import _lib
    FAKE = 0x1000

    def gemm(**kw):
        p = _lib.GemmParams()
        for k in ("a0", "w", "out"):
            setattr(p, k, FAKE)
        for k, v in kw.items():
            setattr(p, k, v)
        return _lib.kernel_info(lib.pv_gemm_conv_kernel_info, p)

    conv = lambda B, s, cin, cout, **kw: gemm(c0=cin, lda0=cin, N=cout, ldc=cout, M=B * s * s, taps=9, batch=B, hin=s, win=s, hout=s, wout=s, stride=1, pad=1, **kw)
    lin = lambda M, K, N, **kw: gemm(c0=K, lda0=K, N=N, ldc=kw.pop("ldc", N), M=M, taps=1, batch=1, hin=1, win=1, hout=M, wout=1, stride=1, pad=1, **kw)
    assert conv(16, 64, 320, 320, colstats=FAKE) == ("big_tile_kernel<true, false, 8, 3, false>", 256)
    assert conv(16, 32, 640, 640, colstats=FAKE) == ("gemm_conv_kernel<5, true, false, true, false, 4>", 512)          # 128 tiles of 256 rows: below the default threshold
    assert conv(16, 32, 640, 640, colstats=FAKE, big_tile_min=128) == ("big_tile_kernel<true, false, 8, 0, false>", 128)   # beside the CFG twin: half the chip
    assert conv(32, 16, 1280, 1280, splitk=2, splitk_ws=FAKE, big_tile_min=256) == ("big_tile_kernel<false, false, 8, 0, false>", 256)
    assert conv(32, 8, 1280, 1280, splitk=4, splitk_ws=FAKE) == ("gemm_conv_kernel<5, true, false, false, false, 4>", 512)
    assert lin(65536, 320, 320) == ("gemm_conv_kernel<5, false, false, false, false, 4>", 1024)
    assert lin(65536, 1280, 320) == ("big_tile_kernel<false, false, 8, 1, false>", 256)
    assert lin(8192, 1280, 1280) == ("gemm_conv_kernel<5, false, false, false, false, 4>", 512)                       # PV_GEMM_BIG128 (off): 128 x 320 tiles
    assert lin(16384, 640, 5120, geglu=1, ldc=2560) == ("big_tile_kernel<false, false, 8, 2, false>", 1280)
    assert lin(1232, 768, 640) == ("gemm_conv_kernel<5, false, false, false, false, 2>", 80)                           # 64-row tiles for small launches
    with pytest.raises(ValueError):
        lin(65536, 300, 320)                                                                                           # K % 64

    def attn(B, n, d, **kw):
        a = _lib.AttnParams(q=FAKE, k=FAKE, v=FAKE, out=FAKE, ldq=24 * d, ldk=24 * d, ldv=24 * d, ldo=8 * d, batch=B, heads=8, nq=n, nk=n, d=d, **kw)
        return _lib.kernel_info(lib.pv_attention_kernel_info, a)
    assert attn(16, 4096, 40) == ("attn8_kernel<497>", 1024) and attn(4, 9216, 40) == ("attn8_kernel<497>", 576)
    assert attn(1, 4096, 40) == ("attn_kernel<40, 2, true>", 256)                                                      # 64 512-query workgroups: below one per CU
    assert attn(16, 4096, 40, causal=1)[0] == "attn_kernel<40, 4, true>"
    assert attn(16, 1024, 80) == ("attn_kernel<80, 2, false>", 1024) and attn(32, 256, 160) == ("attn_kernel<160, 2, false>", 512)
    with pytest.raises(ValueError):
        attn(16, 1024, 48)


def test_splitk_choice_by_wave_quantisation(monkeypatch):
    """``ops.choose_splitk`` (pure host logic): the headline's shapes keep the slice counts of round 5; configs[4]'s per-rank shape gets the counts wave
    quantisation prefers (72 one-per-CU tiles x 3 slices = ONE round of workgroups instead of x 4 = two; 72 tiles of the 128-row kernel x 7 slices = 504 of
    512 workgroup slots instead of x 4 = 288); SPLITK_MAX = 1 (the batch-invariance tests) turns every automatic split off; the caller's choice wins."""
    from photoverse_amd import ops
    conv = lambda b, s: (b, s, s, s, s, 1, 0, 1)
    K9 = 9 * 1280
    # headline (bs = 16): merged plan at batch 32
    assert ops.choose_splitk(32 * 256, 1280, K9, conv_geo=conv(32, 16), big_min=256, big_split2=True) == 2      # 16 x 16 convs: 128 tiles x 2
    assert ops.choose_splitk(32 * 64, 1280, K9, conv_geo=conv(32, 8), big_min=256, big_split2=True) == 4        # 8 x 8 convs: 128 tiles of the 128-row kernel x 4 = 512 slots
    assert ops.choose_splitk(16 * 4096, 320, 9 * 320, conv_geo=conv(16, 64), big_min=128) == 1                  # chip-filling convs: no split
    assert ops.choose_splitk(32 * 64, 1280, 1280) == 1                                                           # short-K Linear: no split
    # configs[4] per-rank shape (bs = 4): merged plan at batch 8
    assert ops.choose_splitk(8 * 576, 1280, K9, conv_geo=conv(8, 24), big_min=256, big_split2=True) == 3        # 72 one-per-CU tiles: 216 workgroups, one round
    assert ops.choose_splitk(8 * 144, 1280, K9, conv_geo=conv(8, 12), big_min=256, big_split2=True) == 7        # 72 tiles x 7 = 504 of 512 slots
    monkeypatch.setattr(ops, "QUANT_SPLITK", False)
    monkeypatch.setattr(ops, "QUANT_SPLITK2", False)
    assert ops.choose_splitk(8 * 576, 1280, K9, conv_geo=conv(8, 24), big_min=256, big_split2=True) == 4        # the round-5 rule
    assert ops.choose_splitk(8 * 144, 1280, K9, conv_geo=conv(8, 12), big_min=256, big_split2=True) == 4
    monkeypatch.setattr(ops, "QUANT_SPLITK", True)
    monkeypatch.setattr(ops, "QUANT_SPLITK2", True)
    # the caller's choice, and split-K off
    assert ops.choose_splitk(8 * 144, 1280, K9, splitk=0, conv_geo=conv(8, 12)) == 1 and ops.choose_splitk(8 * 144, 1280, K9, splitk=3, conv_geo=conv(8, 12)) == 3
    assert ops.choose_splitk(4096, 5120, 640, geglu=True) == 1
    monkeypatch.setattr(ops, "SPLITK_MAX", 1)
    for M, s in ((32 * 256, 16), (32 * 64, 8), (8 * 576, 24), (8 * 144, 12), (2 * 64, 8), (64, 8)):
        assert ops.choose_splitk(M, 1280, K9, conv_geo=conv(M // (s * s), s), big_min=256, big_split2=True) == 1, (M, s)
        assert ops.choose_splitk(M, 1280, 64, conv_geo=None) == 1


def test_struct_layouts_match_header():
    """ctypes mirrors of the parameter structs: field names and order equal the header's."""
    from photoverse_amd import _lib
    header = open(os.path.join(ROOT, "include", "photoverse_hip.h")).read()
    for cname, cls in (("pv_gemm_params", _lib.GemmParams), ("pv_groupnorm_params", _lib.GroupNormParams),
                       ("pv_layernorm_params", _lib.LayerNormParams), ("pv_attn_params", _lib.AttnParams),
                       ("pv_xattn_params", _lib.XAttnParams), ("pv_xattn_fused_params", _lib.XAttnFusedParams),
                       ("pv_xattn_lnq_params", _lib.XAttnLnqParams), ("pv_row_gemm_params", _lib.RowGemmParams),
                       ("pv_attn_bwd_params", _lib.AttnBwdParams), ("pv_groupnorm_bwd_params", _lib.GroupNormBwdParams),
                       ("pv_xattn_bwd_params", _lib.XAttnBwdParams), ("pv_layernorm_bwd_params", _lib.LayerNormBwdParams)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), header, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(",")
            names.append(re.findall(r"(\w+)\s*$", parts[0].strip())[0])
            names += [re.findall(r"(\w+)\s*$", p.strip())[0] for p in parts[1:]]
        assert names == [f[0] for f in cls._fields_], (cname, names)


def test_product_refuses_cpu():
    from photoverse_amd.ops import Recorder
    from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter
    with pytest.raises(RuntimeError, match="no CPU path"):
        Recorder("cpu")
    u = UNet2DConditionModel(**TINY)
    set_visual_cross_attention_adapter(u, (5,))
    with pytest.raises(RuntimeError, match="no CPU path"):
        u(torch.randn(1, 4, 16, 16), torch.tensor(3), encoder_hidden_states=(torch.randn(1, 77, 768), torch.randn(1, 1, 768)))
    with pytest.raises(NotImplementedError):
        u.conv_in  # holders exist ...
        u.mid_block(torch.zeros(1))   # ... but have no eager forward


def test_product_never_imports_oracle():
    """The product package must not reach into oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "photoverse_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
    for f in ("generate.py",):
        assert "oracle" not in open(os.path.join(ROOT, f)).read()


def test_unet_state_dict_names_and_processors():
    from oracle.unet_ref import UNet2DConditionModelRef, set_visual_cross_attention_adapter_ref
    from photoverse_amd.attention_processor import AttnProcessor2_0, PhotoVerseAttnProcessor2_0
    from photoverse_amd.unet import UNet2DConditionModel, set_visual_cross_attention_adapter
    with torch.device("meta"):
        ref = UNet2DConditionModelRef()
        hip = UNet2DConditionModel()
    set_visual_cross_attention_adapter_ref(ref, (5,))
    set_visual_cross_attention_adapter(hip, (5,))
    a, b = ref.state_dict(), hip.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape for k in a)
    assert sum(p.numel() for n, p in hip.named_parameters() if "processor" not in n) == 859_520_964
    procs = hip.attn_processors
    assert len(procs) == 32
    assert sum(isinstance(p, PhotoVerseAttnProcessor2_0) for p in procs.values()) == 16
    assert all(isinstance(p, AttnProcessor2_0) for n, p in procs.items() if n.endswith("attn1.processor"))
    sizes = {n: p.hidden_size for n, p in procs.items() if isinstance(p, PhotoVerseAttnProcessor2_0)}
    assert sizes["down_blocks.0.attentions.0.transformer_blocks.0.attn2.processor"] == 320
    assert sizes["mid_block.attentions.0.transformer_blocks.0.attn2.processor"] == 1280
    assert sizes["up_blocks.3.attentions.2.transformer_blocks.0.attn2.processor"] == 320
    for bad in (dict(fusion_rules=(0.5, 0.6)), dict(fusion_rules=[1 / 3, 2 / 3]), dict(scale=[1.0, 2.0])):
        with pytest.raises(ValueError):
            PhotoVerseAttnProcessor2_0(320, 768, **bad)
    p = PhotoVerseAttnProcessor2_0(320, 768)
    with torch.no_grad():
        assert p.branch_weights() == (1.0, 1.0)
    for seed, exp in ((0.1, (2.0, 0.0)), (0.9, (0.0, 2.0)), (0.5, (1.0, 1.0))):
        p.forced_fusion_seed = seed
        assert p.branch_weights() == exp


def test_scheduler_table_reproduces_oracle_stepping():
    from oracle.scheduler_ref import DPMSolverMultistepRef
    from photoverse_amd.scheduler import DPMSolverMultistepScheduler
    for n in (2, 7, 25, 50):
        s = DPMSolverMultistepScheduler.from_config(DPMSolverMultistepScheduler().config)
        s.set_timesteps(n)
        tab = s.coefficient_table().double()
        r = DPMSolverMultistepRef()
        r.set_timesteps(n)
        assert torch.equal(s.timesteps, r.timesteps) and tab.shape == (n, 8)
        g = torch.Generator().manual_seed(n)
        x = torch.randn(256, generator=g, dtype=torch.float64)
        xr, xp = x.clone(), torch.zeros_like(x)
        for i, t in enumerate(r.timesteps):
            eps = torch.randn(256, generator=g, dtype=torch.float64)
            xr = r.step(eps, t, xr)
            ca, cb, cx, c0, c1 = tab[i, :5]
            x0 = ca * x + cb * eps
            x, xp = cx * x + c0 * x0 + c1 * xp, x0
            assert ((x - xr).norm() / xr.norm()).item() < 1e-6


def test_pack_geglu_is_a_permutation_with_paired_fragments():
    from photoverse_amd.ops import pack_geglu
    n, k = 1280, 8
    w = torch.arange(2 * n, dtype=torch.float32)[:, None].repeat(1, k)
    wp, bp = pack_geglu(w, torch.arange(2 * n, dtype=torch.float32))
    src = wp[:, 0].long()
    assert torch.equal(torch.sort(src).values, torch.arange(2 * n)) and torch.equal(bp.long(), src)
    # every 32 packed rows hold 16 value rows j..j+15 followed by their 16 gate rows n+j..n+j+15
    blk = src.view(-1, 2, 16)
    assert torch.equal(blk[:, 1], blk[:, 0] + n)
    assert torch.equal(blk[:, 0, 0], torch.arange(0, n, 16))


def test_tokenizer_uncond_ids():
    from photoverse_amd.tokenizer import SyntheticCLIPTokenizer
    tok = SyntheticCLIPTokenizer()
    ids = tok([""] * 3, padding="max_length", max_length=tok.model_max_length, return_tensors="pt").input_ids
    assert ids.shape == (3, 77) and ids.dtype == torch.int64
    assert ids[0, 0].item() == 49406 and (ids[:, 1:] == 49407).all()
    ids2 = tok("a photo of a *").input_ids
    assert ids2[0, 0].item() == 49406 and ids2[0, 6].item() == 49407 and ids2.max().item() < 49408


def test_checkpoint_layout_roundtrip_and_lora(tmp_path):
    """save_progress / load_photoverse_model keep the reference layout (modeling_utils.py:13-50)."""
    from photoverse_amd.lora import LoraConfig
    from photoverse_amd.modeling_utils import load_models, load_photoverse_model, save_progress
    tok, te, vae, unet, ie, ia, ta, sch, _ = load_models(None, 1, unet_config=TINY, vision_config=VIS, text_config=TXT, vae_config=VAE, seed=1)
    save_progress(ia, ta, unet, None, str(tmp_path), step=42)
    f = tmp_path / "photoverse_000042.pt"
    assert f.exists()
    ck = torch.load(str(f))
    assert set(ck) == {"image_adapter", "text_adapter", "cross_attention_adapter"}
    keys = list(ck["cross_attention_adapter"])
    assert len(keys) == 4 * 5                       # 4 cross-attn layers x (to_q, to_k, to_v, to_k_ip, to_v_ip)
    assert "mid_block.attentions.0.transformer_blocks.0.attn2.processor.to_k_ip.0.weight" in keys
    assert not any("to_out" in k for k in keys)
    assert set(ck["image_adapter"]) == {f"mapping{p}_{i}.{j}.{w}" for p in ("", "_patch") for i in range(2) for j, ws in
                                        ((0, "wb"), (1, "wb"), (3, "wb"), (4, "wb"), (6, "wb")) for w in ("weight", "bias")}
    # load into a differently seeded model: adapters + cross-attention subset become equal, the rest stays
    tok2, te2, vae2, unet2, ie2, ia2, ta2, sch2, lc = load_models(None, 1, str(f), unet_config=TINY, vision_config=VIS, text_config=TXT, vae_config=VAE, seed=2)
    assert lc is None
    for k in keys:
        assert torch.equal(unet2.state_dict()[k], unet.state_dict()[k])
    assert not torch.equal(unet2.conv_in.weight, unet.conv_in.weight)
    assert all(torch.equal(a, b) for a, b in zip(ia2.state_dict().values(), ia.state_dict().values()))
    # LoRA: peft-style names, config stored, re-injected on load BEFORE the weights
    cfg = LoraConfig(r=4, lora_alpha=8)
    tok3, te3, vae3, unet3, *_rest = load_models(None, 1, use_lora=True, lora_config=cfg, unet_config=TINY, vision_config=VIS, text_config=TXT, vae_config=VAE, seed=1)
    k3 = list(unet3.state_dict())
    base = "mid_block.attentions.0.transformer_blocks.0.attn2.to_q."
    assert base + "base_layer.weight" in k3 and base + "lora_A.default.weight" in k3 and base + "lora_B.default.weight" in k3
    assert unet3.state_dict()[base + "lora_A.default.weight"].shape == (4, 640)
    q = dict(unet3.named_modules())[base[:-1]]
    nn.init.normal_(q.lora_B["default"].weight)
    merged = q.base_layer.weight + 2.0 * q.lora_B["default"].weight @ q.lora_A["default"].weight
    assert torch.allclose(q.weight, merged)
    save_progress(ia, ta, unet3, None, str(tmp_path), lora_config=cfg, optimizer=torch.optim.AdamW(ia.parameters()))
    ck3 = torch.load(str(tmp_path / "photoverse.pt"))
    assert {"optimizer", "lora_config"} <= set(ck3) and ck3["lora_config"]["r"] == 4
    assert any(k.endswith("to_q.lora_B.default.weight") for k in ck3["cross_attention_adapter"])
    *_x, unet4, _ie, _ia, _ta, _s, lc4 = load_models(None, 1, str(tmp_path / "photoverse.pt"), unet_config=TINY, vision_config=VIS, text_config=TXT, vae_config=VAE, seed=5)
    assert lc4 is not None and lc4.r == 4
    assert torch.equal(unet4.state_dict()[base + "lora_B.default.weight"], unet3.state_dict()[base + "lora_B.default.weight"])
    with pytest.raises(AssertionError):
        load_models(None, 1, use_lora=True, unet_config=TINY, vision_config=VIS, text_config=TXT, vae_config=VAE)     # modeling_utils.py:87
    with pytest.raises(FileNotFoundError):
        load_models("runwayml/stable-diffusion-v1-5", 1)                                              # no network here


def _write_synthetic_clip_vocab(d):
    """A small CLIP-style vocabulary (byte alphabet, </w> forms, greedy merges of a tiny corpus): vocab.json + merges.txt."""
    import json
    from photoverse_amd.tokenizer import bytes_to_unicode
    alphabet = list(bytes_to_unicode().values())
    vocab = alphabet + [c + "</w>" for c in alphabet]
    corpus = ("a photo of a person wearing sunglasses on the beach . the quick brown fox's 42 jumps over l'ete don't stop , "
              "photo photos photographer").split()
    words = [tuple(w[:-1]) + (w[-1] + "</w>",) for w in corpus]
    merges = []
    for _ in range(60):
        cnt = {}
        for w in words:
            for a, b in zip(w, w[1:]):
                cnt[(a, b)] = cnt.get((a, b), 0) + 1
        if not cnt:
            break
        best = max(sorted(cnt), key=lambda k: cnt[k])
        merges.append(best)
        nw = []
        for w in words:
            out, i = [], 0
            while i < len(w):
                if i < len(w) - 1 and (w[i], w[i + 1]) == best:
                    out.append(w[i] + w[i + 1]); i += 2
                else:
                    out.append(w[i]); i += 1
            nw.append(tuple(out))
        words = nw
    vocab += [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
    enc = {t: i for i, t in enumerate(vocab)}
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "vocab.json"), "w") as fh:
        json.dump(enc, fh)
    with open(os.path.join(d, "merges.txt"), "w") as fh:
        fh.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    return enc, merges


def test_bpe_tokenizer_matches_installed_transformers(tmp_path):
    """photoverse_amd's CLIP BPE vs the installed transformers CLIPTokenizer built from the SAME vocabulary files."""
    from photoverse_amd.tokenizer import CLIPBPETokenizer, SyntheticCLIPTokenizer, load_tokenizer
    enc, merges = _write_synthetic_clip_vocab(str(tmp_path / "tokenizer"))
    mine = load_tokenizer(str(tmp_path))
    assert isinstance(mine, CLIPBPETokenizer) and isinstance(load_tokenizer(None), SyntheticCLIPTokenizer)
    from transformers import CLIPTokenizer
    hf = CLIPTokenizer(vocab=enc, merges=merges)
    texts = ["a photo of a *", "A Photo of   a person's sunglasses, 42 foxes!", "", "the   beach .", "photographer l'ete don't", "x " * 100]
    for t in texts:
        a = mine(t, padding="max_length", max_length=77, return_tensors="pt").input_ids
        b = hf(t, padding="max_length", truncation=True, max_length=77, return_tensors="pt").input_ids
        assert a.shape == (1, 77) and torch.equal(a, b), t
    batch = mine(texts[:3], padding="max_length", max_length=77, return_tensors="pt").input_ids
    assert batch.shape == (3, 77) and batch[2, 0] == enc["<|startoftext|>"] and (batch[2, 1:] == enc["<|endoftext|>"]).all()


def test_load_models_from_local_hf_layout(tmp_path):
    from safetensors.torch import save_file
    from photoverse_amd.modeling_utils import load_models
    from photoverse_amd.tokenizer import CLIPBPETokenizer
    tok, te, vae, unet, ie, *_ = load_models(None, 1, unet_config=TINY, vision_config=VIS, text_config=TXT, vae_config=VAE, seed=7)
    (tmp_path / "unet").mkdir(); (tmp_path / "text_encoder").mkdir(); (tmp_path / "vae").mkdir()
    clipdir = tmp_path / "clip-vit"                      # the reference pulls the image encoder from a separate repository
    clipdir.mkdir()
    # the published SD-v1.5 VAE file uses the deprecated attention names (query / key / value / proj_attn)
    dep = {".to_q.": ".query.", ".to_k.": ".key.", ".to_v.": ".value.", ".to_out.0.": ".proj_attn."}
    vsd = {}
    for k, v in vae.state_dict().items():
        if ".attentions." in k:
            for new, old in dep.items():
                k = k.replace(new, old)
        vsd[k] = v.contiguous()
    assert "encoder.conv_in.weight" in vsd and "quant_conv.weight" in vsd and any(".proj_attn." in k for k in vsd)
    save_file(vsd, str(tmp_path / "vae" / "diffusion_pytorch_model.safetensors"))
    plain = {k: v.contiguous() for k, v in unet.state_dict().items() if "processor" not in k}
    save_file(plain, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    save_file({k: v.contiguous() for k, v in te.state_dict().items()}, str(tmp_path / "text_encoder" / "model.safetensors"))
    full_clip = {k: v.contiguous() for k, v in ie.state_dict().items()}
    full_clip["text_model.embeddings.token_embedding.weight"] = torch.zeros(4, 4)      # the CLIP repo holds both towers
    full_clip["visual_projection.weight"] = torch.zeros(4, 4)
    save_file(full_clip, str(clipdir / "model.safetensors"))
    kw = dict(unet_config=TINY, vision_config=VIS, text_config=TXT, vae_config=VAE, seed=8)
    # loud, not silent: no tokenizer files / no image encoder where expected
    with pytest.raises(FileNotFoundError, match="tokenizer"):
        load_models(str(tmp_path), 1, image_encoder_path=str(clipdir), **kw)
    _write_synthetic_clip_vocab(str(tmp_path / "tokenizer"))
    with pytest.raises(FileNotFoundError, match="image encoder"):
        load_models(str(tmp_path), 1, **kw)
    tok2, te2, vae2, unet2, ie2, *_ = load_models(str(tmp_path), 1, image_encoder_path=str(clipdir), **kw)
    assert isinstance(tok2, CLIPBPETokenizer)
    assert all(torch.equal(unet2.state_dict()[k], v) for k, v in plain.items())
    assert all(torch.equal(a, b) for a, b in zip(te2.state_dict().values(), te.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(ie2.state_dict().values(), ie.state_dict().values()))
    assert all(torch.equal(a, b) for a, b in zip(vae2.state_dict().values(), vae.state_dict().values()))     # deprecated names mapped
    assert not any(p.requires_grad for p in vae2.parameters())
    assert not any(p.requires_grad for p in unet2.conv_in.parameters())                # frozen (modeling_utils.py:63-66)
    assert all(p.requires_grad for n, p in unet2.named_parameters() if "processor" in n)   # created after the freeze
    # a weight file that misses parameters raises; strict_load=False downgrades to warnings and keeps the seeded init
    partial = {k: v for k, v in plain.items() if "mid_block" not in k}
    save_file(partial, str(tmp_path / "unet" / "diffusion_pytorch_model.safetensors"))
    with pytest.raises(KeyError, match="UNet"):
        load_models(str(tmp_path), 1, image_encoder_path=str(clipdir), **kw)
    with pytest.warns(UserWarning, match="random-init"):
        load_models(str(tmp_path), 1, image_encoder_path=str(clipdir), strict_load=False, **kw)


def test_adapter_default_init_equals_real_reference(golden_dir):
    """Product adapter built under the golden seed holds the REAL reference's weights (checksums from make_golden.py)."""
    from photoverse_amd.adapters import PhotoVerseAdapter
    g = torch.load(os.path.join(golden_dir, "adapter_golden.pt"))
    torch.manual_seed(g["weights_seed"])
    sd = PhotoVerseAdapter(1024, 768, num_tokens=2).state_dict()
    assert set(sd) == set(g["weight_checksums"])
    for k, (s1, s2, head) in g["weight_checksums"].items():
        assert sd[k].double().sum().item() == pytest.approx(s1, rel=1e-12, abs=1e-12) and torch.equal(sd[k].flatten()[:4], head)


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from photoverse_amd.pipeline import gather_latents, shard_batch
    g = torch.Generator().manual_seed(5)
    noise = torch.randn(8, 4, 16, 16, generator=g)          # global batch drawn identically on every rank (infer.py:52-59 semantics)
    sl = shard_batch(8, rank, world)
    local = noise[sl] * 2.0 + 1.0                            # stands for the rank's independent denoise of its shard
    full = gather_latents(local, world)
    ok = torch.equal(full, noise * 2.0 + 1.0)
    q.put((rank, ok, tuple(full.shape)))
    dist.destroy_process_group()


def _gloo_pipeline_worker(rank, world, port, q):
    """PhotoVersePipeline(shard=True, seed=...) under gloo: the host logic of the sharded entry point (example slicing, ONE global
    noise draw sliced per rank, one gather) with the device part (run_inference) replaced by a stand-in that returns a function of
    exactly what it was handed."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import photoverse_amd.infer as infer_mod
    from photoverse_amd.pipeline import PhotoVersePipeline
    from types import SimpleNamespace

    def fake_run_inference(example, *a, noise=None, seed=None, latent_size=64, **kw):
        assert noise is not None and seed is not None
        return noise * 3.0 + example["pixel_values_clip"].mean(dim=(1, 2, 3)).view(-1, 1, 1, 1)

    infer_mod.run_inference = fake_run_inference
    unet = SimpleNamespace(config=SimpleNamespace(in_channels=4))
    pipe = PhotoVersePipeline(None, None, None, unet, None, None, None, None)
    g = torch.Generator().manual_seed(11)
    example = {"pixel_values_clip": torch.randn(6, 3, 8, 8, generator=g), "concept_placeholder_idx": torch.zeros(6, 1, dtype=torch.int64),
               "text": ["x"] * 6}
    out = pipe(example, shard=True, seed=123, latent_size=16)
    ref_noise = torch.randn((6, 4, 16, 16), generator=torch.manual_seed(123))     # the seeded 1-GPU draw (infer.py:52-59)
    expect = ref_noise * 3.0 + example["pixel_values_clip"].mean(dim=(1, 2, 3)).view(-1, 1, 1, 1)
    q.put((rank, bool(torch.equal(out, expect)), tuple(out.shape)))
    dist.destroy_process_group()


def test_sharded_pipeline_seeded_noise_equals_one_rank_draw_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_pipeline_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True, (6, 4, 16, 16)), (1, True, (6, 4, 16, 16))]


def test_bench_self_launches_n_ranks_dry():
    """`python bench.py --gpus N` spawns N ranks itself (no torchrun): distinct ranks / processes, n_gpus == world size;
    with no GPUs it fails cleanly; a mismatching external launcher is refused."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-launch"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 3 and sorted(line["ranks"]) == [0, 1, 2] and sorted(line["local_ranks"]) == [0, 1, 2] and line["distinct_pids"] == 3
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"], capture_output=True, text=True,
                           env=env, timeout=300)
        assert r.returncode != 0 and "only 0 HIP device" in r.stderr and r.stdout.strip() == ""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], capture_output=True, text=True,
                       env=dict(env, RANK="0", WORLD_SIZE="1"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_bench_dry_launch_at_the_real_world_size_and_workload_labels():
    """VERDICT round 4 item 7 (no 8-GPU node exists for the build): `bench.py --gpus 8 --dry-launch` starts 8 distinct rank processes,
    runs the data path's one collective (all_gather_into_tensor of the per-rank shards, gloo) and finds the global batch back in rank
    order; the workload label reads configs[2] for `--gpus 8 --batch 16` and configs[4] for `--gpus 8 --batch 4 --latent 96 --ip-tokens 6`."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}

    def dry(*extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-launch", *extra], capture_output=True, text=True,
                           env=env, timeout=600)
        assert r.returncode == 0, r.stderr
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    a = dry("--batch", "16")
    assert a["n_gpus"] == 8 and sorted(a["ranks"]) == list(range(8)) and sorted(a["local_ranks"]) == list(range(8)) and a["distinct_pids"] == 8
    assert a["gather_in_rank_order"] and a["global_batch"] == 128 and a["workload"].startswith("configs[2]: bs=128")
    b = dry("--batch", "4", "--latent", "96", "--ip-tokens", "6")
    assert b["gather_in_rank_order"] and b["global_batch"] == 32 and b["workload"].startswith("configs[4]: bs=32") and "768x768" in b["workload"] and "P=6" in b["workload"]
    sys.path.insert(0, ROOT)
    import bench
    assert bench.workload_label(16, 64, 1, 7.5, 50, 1).startswith("configs[1]: ")
    assert bench.workload_label(16, 64, 1, 7.5, 20, 1).startswith("configs[1] shape timed over 20 steps")
    assert bench.workload_label(4, 96, 6, 7.5, 50, 1).startswith("configs[4] per-rank shape")
    assert bench.workload_label(8, 64, 1, 7.5, 50, 8).startswith("custom shape")


def test_batch_shard_and_single_gather_gloo_world2():
    import torch.multiprocessing as mp
    from photoverse_amd.pipeline import shard_batch
    assert shard_batch(128, 3, 8) == slice(48, 64)
    with pytest.raises(ValueError):
        shard_batch(10, 0, 4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    assert res == [(0, True, (8, 4, 16, 16)), (1, True, (8, 4, 16, 16))]


def test_ddim_table_reproduces_stepwise_ddim():
    from oracle.scheduler_ref import DDIMRef
    from photoverse_amd.scheduler import DDIMScheduler
    for n in (10, 50):
        s = DDIMScheduler()
        s.set_timesteps(n)
        tab = s.coefficient_table().double()
        r = DDIMRef()
        r.set_timesteps(n)
        assert torch.equal(s.timesteps, r.timesteps) and tab[:, 4].abs().max() == 0
        g = torch.Generator().manual_seed(n)
        x = torch.randn(64, generator=g, dtype=torch.float64)
        xr = x.clone()
        for i, t in enumerate(r.timesteps):
            eps = torch.randn(64, generator=g, dtype=torch.float64)
            xr = r.step(eps, t, xr)
            ca, cb, cx, c0, _ = tab[i, :5]
            x = cx * x + c0 * (ca * x + cb * eps)
            assert ((x - xr).norm() / xr.norm()).item() < 1e-6


def test_image_preprocessing_matches_installed_clip_image_processor():
    """generate.py:57-58 geometry: CLIP pixels = short side 224 (bicubic) + centre crop, NOT a squash; VAE pixels = short side
    512 + centre crop in [-1, 1].  Pinned against the installed transformers CLIPImageProcessor (PIL backend)."""
    import numpy as np
    from PIL import Image
    from transformers import CLIPImageProcessor
    from photoverse_amd.image_utils import clip_image_processor, denormalize, preprocess_image, to_pil
    rng = np.random.default_rng(0)
    for (h, w) in ((300, 200), (250, 640), (224, 224), (1024, 768)):
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        exp = CLIPImageProcessor()(images=img, return_tensors="pt").pixel_values[0]
        got = clip_image_processor(img)
        assert got.shape == (3, 224, 224) and torch.allclose(got, exp, atol=1e-6), (h, w)
        # the 512 path is the same pipeline with mean = std = 0.5 (torchvision Resize + CenterCrop + Normalize(0.5, 0.5));
        # sizes chosen so that torchvision's round() and transformers' floor() crop offsets agree
        proc = CLIPImageProcessor(size={"shortest_edge": 128}, crop_size={"height": 128, "width": 128}, image_mean=[0.5] * 3, image_std=[0.5] * 3)
        exp2 = proc(images=img, return_tensors="pt").pixel_values[0]
        got2 = preprocess_image(img, size=128, interpolation="bicubic")
        assert got2.shape == (3, 128, 128) and got2.min() >= -1 and got2.max() <= 1
        nw, nh = (128, int(128 * h / w)) if w < h else (int(128 * w / h), 128)
        if (nh - 128) % 2 == 0 and (nw - 128) % 2 == 0:
            assert torch.allclose(got2, exp2, atol=1e-6), (h, w)
    # a 2:1 image: the centre crop keeps the middle, a squash would keep the borders
    arr = np.zeros((100, 200, 3), dtype=np.uint8)
    arr[:, :50] = 255
    out = preprocess_image(Image.fromarray(arr), size=64)
    assert out.max() < -0.8                                   # the white left quarter is cropped away (bicubic ringing only)
    back = to_pil(denormalize(torch.full((3, 4, 4), 0.0)))
    assert back.size == (4, 4) and np.asarray(back)[0, 0, 0] == 128


def test_generate_prepare_example_from_image_file(tmp_path):
    """generate.py: an image file -> CLIP pixels (224x224, CLIP mean/std) + VAE pixels in [-1,1] (generate.py:53-61 of the reference)."""
    import argparse
    import importlib.util
    import numpy as np
    from PIL import Image
    from photoverse_amd.tokenizer import SyntheticCLIPTokenizer
    spec = importlib.util.spec_from_file_location("pv_generate", os.path.join(ROOT, "generate.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    rng = np.random.default_rng(0)
    path = tmp_path / "face.png"
    Image.fromarray(rng.integers(0, 256, (300, 200, 3), dtype=np.uint8)).save(path)
    args = argparse.Namespace(num_of_samples=2, text="a photo of a {}", negative_prompt="blurry", synthetic_input=False,
                              input_image_path=str(path), seed=None, latent_size=32)
    ex = gen.prepare_example(args, SyntheticCLIPTokenizer())
    assert ex["pixel_values_clip"].shape == (2, 3, 224, 224) and ex["pixel_values"].shape == (2, 3, 256, 256)
    assert ex["pixel_values"].min() >= -1 and ex["pixel_values"].max() <= 1
    assert torch.equal(ex["pixel_values_clip"][0], ex["pixel_values_clip"][1])
    assert ex["text_input_ids"].shape == (2, 77) and ex["negative_text_input_ids"].shape == (2, 77)
    assert ex["concept_placeholder_idx"].tolist() == [[5], [5]]          # "a photo of a *": 4 words + BOS
    args.synthetic_input = True
    ex2 = gen.prepare_example(args, SyntheticCLIPTokenizer())
    assert ex2["pixel_values"].abs().sum() == 0 and ex2["pixel_values_clip"].shape == (2, 3, 224, 224)


def test_train_cli_rejects_what_it_does_not_support_and_fails_loudly_without_a_gpu():
    """train.py (the training CLI): flags of the reference that this build cannot honour are rejected by the parser, not ignored; without a
    HIP device the run stops with a message (no CPU fallback)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = lambda *a: subprocess.run([sys.executable, os.path.join(root, "train.py"), *a], capture_output=True, text=True, timeout=300)
    for bad, msg in ((["--synthetic_data", "--face_loss", "facenet"], "facenet"),
                     (["--synthetic_data", "--gradient_accumulation_steps", "0"], "gradient_accumulation_steps"),
                     (["--synthetic_data", "--push_to_hub"], "network"), ([], "--data_root_path"),
                     (["--synthetic_data", "--extra_num_tokens", "2"], "image_encoder_layers_idx")):
        r = run(*bad)
        assert r.returncode == 2 and msg in r.stderr, (bad, r.stderr[-300:])
    if not torch.cuda.is_available():
        r = run("--synthetic_data", "--tiny", "--max_train_steps", "1")
        assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout)


def test_train_cli_datasets(tmp_path):
    """The image-folder datasets of the training CLI (datasets/custom.py:44-171): numbered files in order, prompt + placeholder index,
    [-1, 1] VAE pixels and CLIP pixels; with masks the CLIP image is the masked photo cropped to the reference's enlarged, squared box."""
    import importlib.util
    import numpy as np
    from PIL import Image
    from photoverse_amd.tokenizer import SyntheticCLIPTokenizer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pv_train_cli", os.path.join(root, "train.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    os.makedirs(tmp_path / "images"); os.makedirs(tmp_path / "masks")
    rng = np.random.RandomState(0)
    for i in (2, 10, 1):                                   # numeric order, not lexicographic
        Image.fromarray(rng.randint(0, 255, (96, 128, 3), dtype=np.uint8)).save(tmp_path / "images" / f"{i}.png")
        m = np.zeros((48, 64), dtype=np.uint8)
        m[10:30, 20:28] = 255                              # tall box: 20 x 8 (bbox ymax - ymin = 19, xmax - xmin = 7)
        Image.fromarray(m).save(tmp_path / "masks" / f"{i}.png")
    tok = SyntheticCLIPTokenizer()
    ds = cli.ImageFolderDataset(str(tmp_path), tok, size=64)
    assert [os.path.basename(p) for p in ds.image_paths] == ["1.png", "2.png", "10.png"] and len(ds) == 3
    ex = ds[0]
    assert ex["pixel_values"].shape == (3, 64, 64) and ex["pixel_values_clip"].shape == (3, 224, 224) and -1 <= ex["pixel_values"].min() and ex["pixel_values"].max() <= 1
    assert ex["text_input_ids"].shape == (77,) and int(ex["concept_placeholder_idx"]) == 4          # "a photo of *": BOS + 3 words
    # the reference's box: bbox rows 10..29, cols 20..27 -> +-15 %: rows 7..31, cols 18..28 -> taller than wide (24 > 10): cols widened by 12 each side
    m = np.zeros((48, 64), dtype=np.uint8); m[10:30, 20:28] = 255
    assert cli.crop_box_of_mask(m) == (7, 31, 6, 40)
    wide = np.zeros((48, 64), dtype=np.uint8); wide[20:24, 2:60] = 1
    assert cli.crop_box_of_mask(wide) == (0, 48, 0, 64)                                            # clipped to the image on every side
    dm = cli.MaskedImageFolderDataset(str(tmp_path), tok, size=64)
    exm = dm[1]
    assert exm["pixel_values_clip"].shape == (3, 224, 224) and torch.equal(exm["pixel_values"], ds[1]["pixel_values"])
    assert not torch.equal(exm["pixel_values_clip"], ds[1]["pixel_values_clip"])
    batch = cli.collate([ds[0], ds[1]])
    assert batch["pixel_values"].shape == (2, 3, 64, 64) and batch["concept_placeholder_idx"].shape == (2, 1)


def _reducer_rank(rank, world, port, q):
    import torch.distributed as dist
    from photoverse_amd.train import GradientReducer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in ((3, 4), (7,), (2, 2, 2))] + [torch.nn.Parameter(torch.zeros(5), requires_grad=False)]
    grads = [[torch.randn(p.shape, generator=g) for p in params[:3]] for _ in range(world)]
    for p, x in zip(params, grads[rank]):
        p.grad = x.clone()
    params[1].grad = None if rank == 1 else params[1].grad                  # a parameter outside one rank's step adds nothing from it
    red = GradientReducer(params)
    n = red()
    ok = n == world and len(red.params) == 3 and red.flat.numel() == 12 + 7 + 8
    ok &= torch.equal(params[0].grad, grads[0][0] + grads[1][0]) and torch.equal(params[2].grad, grads[0][2] + grads[1][2])
    ok &= params[0].grad.data_ptr() == red.flat.data_ptr()                  # gradients are views of the one bucket
    if rank == 0:
        ok &= torch.equal(params[1].grad, grads[0][1])
    else:
        ok &= params[1].grad is None
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_gradient_reducer_two_ranks_gloo():
    """World-size-2 gloo run of the data-parallel gradient sum (photoverse_amd.train.GradientReducer): one flat bucket, one all_reduce."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_reducer_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert res == {0: True, 1: True}
    from photoverse_amd.train import GradientReducer
    lone = [torch.nn.Parameter(torch.zeros(3))]
    lone[0].grad = torch.ones(3)
    assert GradientReducer(lone)() == 1 and torch.equal(lone[0].grad, torch.ones(3))       # no process group: nothing to do


def test_every_golden_fixture_is_plain_data(golden_dir):
    """fixtures are tensors / dicts / lists / scalars only - they load under ``weights_only=True`` (no pickled code can run
    on load); the scripts that execute reference code (oracle/ref_exec.py, oracle/make_*golden.py) run in the build container only."""
    import glob
    files = sorted(glob.glob(os.path.join(golden_dir, "*.pt")))
    assert len(files) >= 20
    for f in files:
        torch.load(f, map_location="cpu", weights_only=True)
