"""ctypes binding of ``libphotoverse_hip.so`` (C-ABI declared in ``include/photoverse_hip.h``).

The product path has NO fallback: if the library is missing or fails to load, importing any op
raises.  ``photoverse_amd.build.build_lib()`` (or ``__graft_entry__.build()``) creates it.
"""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB

#: diagnostics only (tools/ab_lib_bench.sh: same-box A/B of two builds of the library): another build of the SAME sources' library, never a fallback
LIB = os.environ.get("PV_HIP_LIB") or LIB

c_void_p, c_int, c_float, c_int64 = C.c_void_p, C.c_int32, C.c_float, C.c_int64


class GemmParams(C.Structure):
    _fields_ = [("a0", c_void_p), ("a1", c_void_p), ("c0", c_int), ("c1", c_int), ("lda0", c_int), ("lda1", c_int),
                ("w", c_void_p), ("bias", c_void_p), ("rowadd", c_void_p), ("rowadd_ld", c_int), ("residual", c_void_p),
                ("ldr", c_int), ("out", c_void_p), ("ldc", c_int), ("M", c_int), ("N", c_int), ("taps", c_int),
                ("batch", c_int), ("hin", c_int), ("win", c_int), ("hout", c_int), ("wout", c_int), ("stride", c_int),
                ("upsample", c_int), ("pad", c_int), ("act", c_int), ("out_f32", c_int), ("geglu", c_int),
                ("splitk", c_int), ("splitk_ws", c_void_p), ("colstats", c_void_p), ("ln_rowsum", c_void_p), ("ln_eps", c_float), ("big_tile_min", c_int), ("a_norm", c_void_p), ("a_norm_act", c_int)]


class GroupNormParams(C.Structure):
    _fields_ = [("x0", c_void_p), ("x1", c_void_p), ("c0", c_int), ("c1", c_int), ("ld0", c_int), ("ld1", c_int),
                ("batch", c_int), ("hw", c_int), ("groups", c_int), ("splits", c_int), ("partial", c_void_p),
                ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float), ("act", c_int), ("y", c_void_p),
                ("colstats0", c_void_p), ("colstats1", c_void_p)]


class LayerNormParams(C.Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_int), ("y", c_void_p), ("ldy", c_int), ("gamma", c_void_p), ("beta", c_void_p),
                ("rows", c_int), ("cols", c_int), ("eps", c_float), ("act", c_int)]


class AttnParams(C.Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("ldq", c_int), ("ldk", c_int), ("ldv", c_int),
                ("out", c_void_p), ("ldo", c_int), ("batch", c_int), ("heads", c_int), ("nq", c_int), ("nk", c_int),
                ("d", c_int), ("causal", c_int), ("lse", c_void_p)]


class AttnBwdParams(C.Structure):
    _fields_ = [("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("ldq", c_int), ("ldk", c_int), ("ldv", c_int),
                ("out", c_void_p), ("ldo", c_int), ("dout", c_void_p), ("lddo", c_int), ("lse", c_void_p), ("delta", c_void_p), ("qs", c_void_p), ("ldqs", c_int),
                ("dq", c_void_p), ("dk", c_void_p), ("dv", c_void_p), ("lddq", c_int), ("lddk", c_int), ("lddv", c_int),
                ("batch", c_int), ("heads", c_int), ("nq", c_int), ("nk", c_int), ("d", c_int), ("causal", c_int), ("ws", c_void_p), ("ws_bytes", C.c_int64)]


class GroupNormBwdParams(C.Structure):
    _fields_ = [("x0", c_void_p), ("x1", c_void_p), ("c0", c_int), ("c1", c_int), ("ld0", c_int), ("ld1", c_int),
                ("batch", c_int), ("hw", c_int), ("groups", c_int), ("splits", c_int), ("stats", c_void_p), ("stats_stride", c_int),
                ("gamma", c_void_p), ("beta", c_void_p), ("act", c_int), ("dy", c_void_p), ("ld_dy", c_int),
                ("partial", c_void_p), ("sums", c_void_p),
                ("dx0", c_void_p), ("ld_dx0", c_int), ("add0", c_void_p), ("ld_add0", c_int),
                ("dx1", c_void_p), ("ld_dx1", c_int), ("add1", c_void_p), ("ld_add1", c_int)]


class XAttnParams(C.Structure):
    _fields_ = [("q", c_void_p), ("ldq", c_int), ("kt", c_void_p), ("vt", c_void_p), ("ldkt", c_int), ("ldvt", c_int),
                ("kip", c_void_p), ("vip", c_void_p), ("ldkip", c_int), ("ldvip", c_int), ("out", c_void_p), ("ldo", c_int),
                ("vnorm", c_void_p), ("batch", c_int), ("heads", c_int), ("nq", c_int), ("nt", c_int), ("nip", c_int),
                ("d", c_int), ("w_text", c_float), ("w_ip", c_float), ("fusion", c_void_p)]


class XAttnFusedParams(C.Structure):
    _fields_ = [("hs", c_void_p), ("ld_hs", c_int), ("ln", c_int), ("ln_eps", c_float),
                ("wq", c_void_p), ("q_bias", c_void_p), ("wo", c_void_p), ("bias_o", c_void_p), ("kimg", c_void_p), ("vimg", c_void_p),
                ("out", c_void_p), ("ld_out", c_int), ("batch", c_int), ("nq", c_int), ("heads", c_int), ("d", c_int),
                ("nt", c_int), ("nip", c_int), ("w_text", c_float), ("w_ip", c_float), ("fusion", c_void_p), ("rows_per_workgroup", c_int)]


class XAttnLnqParams(C.Structure):
    _fields_ = [("hs", c_void_p), ("ld_hs", c_int), ("ln", c_int), ("ln_eps", c_float), ("wq", c_void_p), ("q_bias", c_void_p), ("wq_rowsum", c_void_p),
                ("kt", c_void_p), ("vt", c_void_p), ("ldkt", c_int), ("ldvt", c_int), ("kip", c_void_p), ("vip", c_void_p), ("ldkip", c_int), ("ldvip", c_int),
                ("out", c_void_p), ("ldo", c_int), ("vnorm", c_void_p), ("batch", c_int), ("nq", c_int), ("heads", c_int), ("d", c_int), ("nt", c_int),
                ("nip", c_int), ("w_text", c_float), ("w_ip", c_float), ("fusion", c_void_p)]


class RowGemmParams(C.Structure):
    _fields_ = [("x", c_void_p), ("ld_x", c_int), ("M", c_int), ("K", c_int), ("N", c_int), ("w", c_void_p), ("bias", c_void_p),
                ("ln", c_int), ("ln_eps", c_float), ("geglu", c_int), ("out", c_void_p), ("ld_out", c_int), ("x_norm", c_void_p), ("rows_per_image", c_int)]


class XAttnBwdParams(C.Structure):
    _fields_ = [("q", c_void_p), ("ldq", c_int), ("kt", c_void_p), ("vt", c_void_p), ("ldkt", c_int), ("ldvt", c_int),
                ("kip", c_void_p), ("vip", c_void_p), ("ldkip", c_int), ("ldvip", c_int), ("dout", c_void_p), ("lddo", c_int),
                ("dq", c_void_p), ("lddq", c_int), ("partial", c_void_p), ("stats", c_void_p), ("dkt", c_void_p), ("dvt", c_void_p), ("dkip", c_void_p),
                ("dvip", c_void_p), ("ld_dt", c_int), ("ld_di", c_int), ("batch", c_int), ("heads", c_int), ("nq", c_int), ("nt", c_int), ("nip", c_int), ("d", c_int),
                ("w_text", c_float), ("w_ip", c_float), ("fusion", c_void_p), ("out_scale", c_float), ("vnorm_coef", c_float), ("vnorm_grad", c_void_p)]


class LayerNormBwdParams(C.Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_int), ("dy", c_void_p), ("lddy", c_int), ("dx", c_void_p), ("lddx", c_int),
                ("gamma", c_void_p), ("beta", c_void_p), ("dgb_partial", c_void_p), ("rows", c_int), ("cols", c_int), ("eps", c_float),
                ("act", c_int), ("dy_group", c_int), ("dy_skip", c_int), ("dy_scale", c_float), ("rows_per_wave", c_int), ("add", c_void_p), ("ldadd", c_int)]


#: every symbol ``include/photoverse_hip.h`` declares: name -> (restype, argtypes)
SIGNATURES = {
    "pv_abi_version": (c_int, []),
    "pv_device_count": (c_int, []),
    "pv_gemm_conv": (c_int, [C.POINTER(GemmParams), c_void_p]),
    "pv_groupnorm_stats": (c_int, [C.POINTER(GroupNormParams), c_void_p]),
    "pv_groupnorm_stats_from_colstats": (c_int, [C.POINTER(GroupNormParams), c_void_p]),
    "pv_groupnorm_scale_shift": (c_int, [C.POINTER(GroupNormParams), c_void_p, c_void_p]),
    "pv_groupnorm_apply": (c_int, [C.POINTER(GroupNormParams), c_void_p]),
    "pv_layernorm": (c_int, [C.POINTER(LayerNormParams), c_void_p]),
    "pv_attention": (c_int, [C.POINTER(AttnParams), c_void_p]),
    "pv_cross_attention": (c_int, [C.POINTER(XAttnParams), c_void_p]),
    "pv_cross_attention_fused": (c_int, [C.POINTER(XAttnFusedParams), c_void_p]),
    "pv_cross_attention_lnq": (c_int, [C.POINTER(XAttnLnqParams), c_void_p]),
    "pv_xattn_pack_kv": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_xattn_fused_wo_slot": (c_int, [c_int]),
    "pv_row_gemm": (c_int, [C.POINTER(RowGemmParams), c_void_p]),
    "pv_cross_attention_backward": (c_int, [C.POINTER(XAttnBwdParams), c_void_p]),
    "pv_transpose_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "pv_layernorm_backward": (c_int, [C.POINTER(LayerNormBwdParams), c_void_p]),
    "pv_reduce_blocks": (c_int, [c_void_p, c_int, c_int64, c_float, c_void_p, c_void_p]),
    "pv_colsum_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "pv_attention_backward": (c_int, [C.POINTER(AttnBwdParams), c_void_p]),
    "pv_groupnorm_backward": (c_int, [C.POINTER(GroupNormBwdParams), c_void_p]),
    "pv_geglu_backward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pv_act_backward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_add_rows_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pv_dilate2x": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_pool2x_sum": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_dropout_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p, c_int, c_int, c_void_p]),
    "pv_col_affine_f16": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pv_prelu_f16": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pv_maxpool2x2": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_gray_resize": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "pv_gray_resize_backward": (c_int, [c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "pv_cosine_embedding_loss": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "pv_wgrad_tn": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "pv_softmax_rows_backward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "pv_clamp_mask_f32": (c_int, [c_void_p, c_void_p, c_float, c_float, c_void_p, c_int64, c_void_p]),
    "pv_act_forward": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_sign_f32": (c_int, [c_void_p, c_float, c_void_p, c_int64, c_void_p]),
    "pv_gather_rows_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "pv_reduce_sumsq": (c_int, [c_void_p, c_int64, c_float, c_void_p, c_int, c_void_p, c_void_p]),
    "pv_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "pv_sumsq_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "pv_clip_coef_groups": (c_int, [c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "pv_adamw_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "pv_geglu": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pv_timestep_embedding": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "pv_conv_out": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_cfg_dpm_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int64, c_void_p]),
    "pv_step_advance": (c_int, [c_void_p, c_void_p]),
    "pv_fusion_draw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_int, c_void_p]),
    "pv_cast_f32_to_f16": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "pv_cast_f16_to_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "pv_rows_mean": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_softmax_rows": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "pv_pointwise_nchw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_affine_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "pv_posterior_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "pv_reduce_mean": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_int, c_void_p, c_void_p]),
    "pv_clamp_f32": (c_int, [c_void_p, c_float, c_float, c_int64, c_void_p]),
    "pv_im2col3x3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "pv_clip_vision_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pv_clip_text_embed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "pv_gemm_conv_kernel_info": (c_int, [C.POINTER(GemmParams), C.c_char_p, c_int, C.POINTER(c_int64)]),
    "pv_attention_kernel_info": (c_int, [C.POINTER(AttnParams), C.c_char_p, c_int, C.POINTER(c_int64)]),
}


def kernel_info(fn, params):
    """(symbol, workgroups) of the launch ``fn`` (pv_gemm_conv_kernel_info / pv_attention_kernel_info) would make for ``params``: the library's own
    dispatch rule, asked - not restated - by the host side for its launch tags."""
    buf = C.create_string_buffer(160)
    wgs = c_int64(0)
    rc = fn(C.byref(params), buf, len(buf), C.byref(wgs))
    if rc != 0:
        raise ValueError(f"{fn.__name__ if hasattr(fn, '__name__') else 'kernel_info'}: the library rejects this parameter block (hipError {rc})")
    return buf.value.decode(), int(wgs.value)

ABI_VERSION = 17
_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def load():
    """Load (once) and type the shared library.  Raises ``HipExtensionMissing`` - never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: PyTorch-ROCm bundles its own libamdhip64.so; a library dlopen-ed before it binds to /opt/rocm's copy, the process ends up with two HIP
    # runtimes and every launch from this one fails with hipErrorNoDevice (100) - seen with `build(); smoke()` in one process
    import torch  # noqa: F401
    if not os.path.exists(LIB):
        raise HipExtensionMissing(
            f"{LIB} not found: build it with `python -m photoverse_amd.build` (needs hipcc). "
            "photoverse_amd has no CPU / eager fallback.")
    try:
        lib = C.CDLL(LIB)
    except OSError as e:  # pragma: no cover
        raise HipExtensionMissing(f"cannot load {LIB}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if lib.pv_abi_version() != ABI_VERSION:
        raise HipExtensionMissing(f"{LIB} has ABI {lib.pv_abi_version()}, expected {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib
