"""Host-side launch recorder over the C-ABI of ``libphotoverse_hip.so``.

A ``Recorder`` turns high-level op calls on torch device tensors into a flat list of
``(c_function, params_struct)`` launches with static buffers.  ``run()`` enqueues the list on the
current HIP stream; because every launch is allocation-free and sync-free the same list can be
captured into a HIP graph (``torch.cuda.CUDAGraph``) and replayed.

torch is used for device memory and streams only; every arithmetic op is a HIP kernel.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import torch

from . import _lib
from ._lib import kernel_info as _kernel_info, load
from ._lib import (AttnBwdParams, AttnParams, GemmParams, GroupNormBwdParams, GroupNormParams, LayerNormBwdParams, LayerNormParams,
                   XAttnBwdParams, XAttnFusedParams, XAttnLnqParams, XAttnParams)

ACT_NONE, ACT_SILU, ACT_QUICK_GELU, ACT_LEAKY_RELU, ACT_GELU = 0, 1, 2, 3, 4

import os as _os
#: split-K heuristic: split until about SPLITK_TARGET workgroups exist (2 per CU), at most SPLITK_MAX ways
SPLITK_MAX = int(_os.environ.get("PV_SPLITK_MAX", "4"))      # same-box sweep (bench, two branches): 4 -> 28.09, 8 -> 27.76, 3 -> 27.91, 2 -> 27.41 steps/s
SPLITK_TARGET = int(_os.environ.get("PV_SPLITK_TARGET", "512"))
#: K slices of a 3x3 conv that reaches one 256 x 320 tile per CU only with split-K (16 x 16 level: 64 tiles); 0 / 1 = keep the 128-row kernel there
BIG_SPLITK = int(_os.environ.get("PV_CONV_BIG_SPLITK", "4"))
#: choose the slice count of such a conv by wave quantisation (rounds of workgroups x K-steps per slice) instead of always BIG_SPLITK; 0 = the round-5 rule
QUANT_SPLITK = _os.environ.get("PV_QUANT_SPLITK", "1") != "0"
QUANT_SPLITK2 = _os.environ.get("PV_QUANT_SPLITK2", "1") != "0"      # the same idea for the 128-row kernel's slice count: +0.85 % at configs[4]'s per-rank shape, level at the headline


def choose_splitk(M: int, N: int, kdim: int, *, geglu: bool = False, splitk=None, conv_geo=None, big_min: int = 256, big_split2: bool = False) -> int:
    """K-slices of a ``Recorder.gemm`` launch (pure host logic; ``splitk``: the caller's choice, None = automatic, 0 = off).  ``conv_geo`` =
    (batch, hin, win, hout, wout, stride, upsample, pad) of a 3x3 conv, None for a Linear layer.

    Automatic rule: split the layers whose 128-row tiles cannot fill the chip (8 x 8 / 16 x 16 levels of the UNet) until about SPLITK_TARGET workgroups
    exist, at most SPLITK_MAX ways (SPLITK_MAX = 1 turns every automatic split off); 3x3 convs that reach one 256 x 320 tile per CU only with split-K take
    the one-per-CU tile with the slice count wave quantisation prefers."""
    auto = splitk is None
    bn = 128 if (geglu or N % 160) else 160
    tiles = ((M + 127) // 128) * (N // bn)
    splitk = 1 if (geglu or splitk == 0) else (splitk or max(1, min(SPLITK_MAX, SPLITK_TARGET // tiles, (kdim // 64) // 16)))
    if auto and QUANT_SPLITK2 and splitk > 1 and splitk == SPLITK_MAX and tiles * splitk * 4 < SPLITK_TARGET * 3:      # never where split-K is off (SPLITK_MAX = 1)
        # the cap, not the chip, stopped the split (configs[4]'s per-rank shape: 72 tiles of the 12 x 12 level x 4 slices = 288 of 512 workgroup
        # slots): more slices while they still fit ONE round, fewest (K-steps per slice + a slab term)
        cands = [k for k in range(splitk, 9) if (kdim // 64) // k >= 16 and tiles * k <= SPLITK_TARGET]
        if cands:
            splitk = min(cands, key=lambda k: (-(-(kdim // 64) // k) + 2 * k, k))
    # pv_convbig.hip's 256 x 320 tile on the 16 x 16 level: 64 tiles x BIG_SPLITK K-slices = one workgroup per CU (the 128-row kernel runs
    # these convs as 256 tiles x 2 slices)
    geo = conv_geo
    up = 2 if (geo is not None and geo[6]) else 1
    big_shape = (geo is not None and big_min > 0 and geo[5] == 1 and geo[7] == 1 and (geo[1] * up, geo[2] * up) == tuple(geo[3:5]) and N % 320 == 0)
    tiles256 = ((M + 255) // 256) * (N // 320) if big_shape else 0
    if (big_shape and auto and 1 < BIG_SPLITK <= SPLITK_MAX and tiles256 * (BIG_SPLITK // 2) < big_min <= tiles256 * BIG_SPLITK
            and (kdim // 64) // BIG_SPLITK >= 16):     # only where it takes ALL the slices to fill the chip (32 x 32 level, 128 tiles: measured slower)
        # ... and of the slice counts that do, the one with the fewest (rounds of workgroups) x (K-steps per slice): at 72 tiles (configs[4]'s per-rank
        # shape, 24 x 24 level of the merged plan) 4 slices are 288 workgroups = two rounds of 45 stages, 3 slices 216 = ONE round of 60
        nk = kdim // 32
        cands = [k for k in range(2, BIG_SPLITK + 1) if (kdim // 64) // k >= 16 and tiles256 * k * 5 >= big_min * 4]
        splitk = min(cands, key=lambda k: (-(-tiles256 * k // big_min) * -(-nk // k), k)) if (cands and QUANT_SPLITK) else BIG_SPLITK
    elif (big_shape and auto and big_split2 and 2 <= SPLITK_MAX and tiles256 < big_min <= tiles256 * 2 and (kdim // 64) // 2 >= 16):
        # the merged low-resolution plan's 16 x 16 convs (batch 2B: 128 tiles): two K-slices on the one-per-CU tile instead of 512 unsplit
        # 128-row workgroups: +0.35 % of a step same-box; the same rule on plans that do not run alone (training, --one-stream) loses 0.4 %
        splitk = 2
    return splitk


class HipLaunchError(RuntimeError):
    pass


def require_cuda(t: torch.Tensor, what: str = "tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"photoverse_amd: {what} must live on a HIP device (got {t.device}); there is no CPU path")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _rows(t: torch.Tensor) -> Tuple[int, int]:
    """(row stride in elements, columns) of a 2-D row view with unit column stride."""
    assert t.dim() == 2 and (t.shape[1] == 1 or t.stride(1) == 1), (t.shape, t.stride())
    return t.stride(0), t.shape[1]


def pack_geglu(weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """Reorder GEGLU ``proj`` rows so that every 128-row weight tile holds, per wave half, alternating
    16-row fragments [value | gate] of the same 16 output columns (see pv_gemm.hip, GEGLU epilogue).
    weight [2*n, k] -> packed [2*n, k]; packed row p maps to source row:
        t = p // 128; wn = (p % 128) // 64; ni = (p % 64) // 16; e = p % 16
        j = t*64 + wn*32 + (ni//2)*16 + e;  src = j + (ni & 1) * n
    """
    n2 = weight.shape[0]
    n = n2 // 2
    assert n2 % 128 == 0
    p = torch.arange(n2, device=weight.device)
    t, wn, ni, e = p // 128, (p % 128) // 64, (p % 64) // 16, p % 16
    src = t * 64 + wn * 32 + (ni // 2) * 16 + e + (ni & 1) * n
    return weight[src].contiguous(), (None if bias is None else bias[src].contiguous())


def pack_geglu_rows(weight: torch.Tensor, bias: Optional[torch.Tensor]):
    """GEGLU ``proj`` rows in the order ``pv_row_gemm`` wants them: per 160-row chunk c, fragment 2q = the 16 VALUE rows of output columns
    80 c + 16 q .. + 15, fragment 2q + 1 = their GATE rows (weight [2 n, k] with value rows first, as diffusers' GEGLU.chunk(2) reads them)."""
    n2 = weight.shape[0]
    n = n2 // 2
    assert n2 % 320 == 0
    p = torch.arange(n2, device=weight.device)
    c, i, e = p // 160, (p % 160) // 16, p % 16
    src = 80 * c + 16 * (i // 2) + e + (i & 1) * n
    return weight[src].contiguous(), (None if bias is None else bias[src].contiguous())


_NO_COLSTATS = bool(os.environ.get("PV_NO_COLSTATS"))   # A/B switch: GroupNorm statistics by a pass over the tensor


class Recorder:
    def __init__(self, device: torch.device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("photoverse_amd: a HIP device is required; there is no CPU path")
        self.calls: List[tuple] = []
        self.tags: List[tuple] = []        # per call: (kernel name, algorithmic flops, algorithmic bytes)
        self.roles: List[Optional[str]] = []   # per call: the role label active when it was recorded (``self.role``; measurement only)
        self.role: Optional[str] = None
        self.keep: List[object] = []       # tensors / structs referenced by raw pointer
        self.colstats: dict = {}           # (data_ptr, rows, cols) of a GEMM output -> its epilogue column statistics
        self.bytes_allocated = 0
        #: minimum number of 256-row tiles for pv_convbig.hip's tile (pv_gemm_params.big_tile_min): 256 = one workgroup per CU; a plan that runs beside
        #: another one on a second stream (the two CFG forwards) uses 128; env PV_CONV_BIG overrides (0 = never)
        self.big_min = int(os.environ["PV_CONV_BIG"]) if "PV_CONV_BIG" in os.environ else 256
        self.big_split2 = False          # set by the plan that runs alone on the chip at batch 2B (UNetEngine(segment="mid"))

    # ------------------------------------------------------------------ memory
    def empty(self, shape, dtype=torch.float16) -> torch.Tensor:
        t = torch.empty(shape, dtype=dtype, device=self.device)
        self.bytes_allocated += t.numel() * t.element_size()
        self.keep.append(t)
        return t

    def hold(self, t):
        self.keep.append(t)
        return t

    # ------------------------------------------------------------------ execution
    def _add(self, fn, *args, tag=None):
        self.keep.extend(a for a in args if isinstance(a, C.Structure))
        self.calls.append((fn, tuple(C.byref(a) if isinstance(a, C.Structure) else a for a in args)))
        self.tags.append(tag if tag is not None else (fn.__name__, 0, 0))
        self.roles.append(self.role)

    def subset(self, pred) -> "Recorder":
        """A recorder sharing this one's buffers that replays only the calls whose tag satisfies ``pred``."""
        r = Recorder.__new__(Recorder)
        r.lib, r.device, r.keep, r.bytes_allocated, r.colstats, r.big_min = self.lib, self.device, self.keep, 0, self.colstats, self.big_min
        r.big_split2 = self.big_split2
        sel = [i for i, t in enumerate(self.tags) if pred(t)]
        r.calls = [self.calls[i] for i in sel]
        r.tags = [self.tags[i] for i in sel]
        r.roles, r.role = [self.roles[i] for i in sel], None
        return r

    @staticmethod
    def concat(*recs: "Recorder") -> "Recorder":
        """A recorder that replays the calls of ``recs`` one after the other (shares their buffers): introspection / timing of a plan that was
        recorded in several segments."""
        r = recs[0].subset(lambda t: False)
        r.keep = list(recs)
        for x in recs:
            r.calls += x.calls
            r.tags += x.tags
            r.roles += x.roles
        return r

    def subset_role(self, role: str) -> "Recorder":
        """The calls recorded under ``self.role == role`` (e.g. the attn2 branch of one channel width), for per-branch timing."""
        r = self.subset(lambda t: True)
        sel = [i for i, ro in enumerate(self.roles) if ro == role]
        r.calls, r.tags, r.roles = [self.calls[i] for i in sel], [self.tags[i] for i in sel], [self.roles[i] for i in sel]
        return r

    def run(self, stream: Optional[int] = None):
        s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        for fn, args in self.calls:
            rc = fn(*args, s)
            if rc != 0:
                raise HipLaunchError(f"{fn.__name__} failed with hipError {rc}")

    def __len__(self):
        return len(self.calls)

    # ------------------------------------------------------------------ ops
    def gemm(self, a: torch.Tensor, w: torch.Tensor, *, a1: Optional[torch.Tensor] = None, bias=None, rowadd=None,
             rowadd_ld: int = 0, rows_per_image: Optional[int] = None, residual=None, out=None, act=ACT_NONE,
             out_f32=False, geglu=False, conv: Optional[dict] = None, splitk: Optional[int] = None,
             colstats: bool = False, colstats_out: Optional[torch.Tensor] = None, ln_gamma=None, ln_beta=None, ln_eps: float = 1e-5,
             a_norm: Optional[torch.Tensor] = None, a_norm_act: int = ACT_NONE) -> torch.Tensor:
        """out = epilogue(A @ W^T).  ``a`` (and ``a1``): 2-D fp16 row views; ``w``: fp16 [N, taps*Cin].
        ``colstats``: the output feeds a GroupNorm - let the epilogue leave its per-column (sum, sum of squares) behind so that
        ``groupnorm`` needs no statistics pass over the tensor (ignored where the epilogue cannot: fp32, GEGLU; with split-K the
        reduce launch produces them).
        ``ln_gamma`` / ``ln_beta``: ``out = epilogue(LayerNorm(a) @ W^T + bias)`` in ONE launch (``gemm_ln_supported`` says where): the affine part is
        folded here, at plan-build time (gamma scales the columns of w - for ``geglu`` BEFORE the caller's ``pack_geglu``, so pass the packed weight
        of an already scaled matrix via ``fold_layernorm`` -, w . beta joins the bias), the kernel normalises through its epilogue.
        ``a_norm`` (``groupnorm_table``'s result) / ``a_norm_act``: ``out = epilogue(conv(act(GroupNorm(a | a1))))`` in ONE launch on the raw tensors
        (``gn_conv_supported`` says where): the 3x3 conv normalises its LDS-resident input patch in place."""
        lda0, c0 = _rows(a)
        lda1, c1 = _rows(a1) if a1 is not None else (0, 0)
        taps = 9 if conv is not None else 1
        N = w.shape[0]
        assert w.shape[1] == taps * (c0 + c1) and w.is_contiguous() and w.dtype == torch.float16, (w.shape, taps, c0, c1)
        if conv is not None:
            M = conv["batch"] * conv["hout"] * conv["wout"]
            geo = (conv["batch"], conv["hin"], conv["win"], conv["hout"], conv["wout"], conv.get("stride", 1), conv.get("upsample", 0), conv.get("pad", 1))
            assert geo[7] == 1 or (geo[7] == 0 and geo[5] == 2 and not geo[6]), "pad=0 is the stride-2 VAE-encoder downsample only"
        else:
            M = a.shape[0]
            geo = (1, 1, 1, rows_per_image or M, 1, 1, 0, 1)
        n_out = N // 2 if geglu else N
        if out is None:
            out = self.empty((M, n_out), torch.float32 if out_f32 else torch.float16)
        ldc, oc = _rows(out)
        assert oc == n_out and out.shape[0] == M
        ldr = 0
        if residual is not None:
            ldr, rc = _rows(residual)
            assert rc == n_out and residual.shape[0] == M
        kdim = taps * (c0 + c1)
        big_min = self.big_min
        splitk = choose_splitk(M, N, kdim, geglu=geglu, splitk=splitk, conv_geo=geo if conv is not None else None, big_min=big_min, big_split2=self.big_split2)
        ws = self.empty((splitk, M, N), torch.float32) if splitk > 1 else None
        cs = None
        key = (out.data_ptr(), M, n_out)
        if colstats and not geglu and not out_f32 and ldc == n_out and not _NO_COLSTATS:
            # ``colstats_out``: caller-owned statistics buffer (an output written in halves by two plans: pipeline.DenoiseLoop's low-resolution merge)
            cs = self.colstats[key] = colstats_out if colstats_out is not None else self.empty(((M + 63) // 64, 2, N), torch.float32)
            assert cs.shape == ((M + 63) // 64, 2, N) and cs.dtype == torch.float32 and cs.is_contiguous()
        else:
            if colstats_out is not None and not _NO_COLSTATS:
                raise ValueError("gemm(colstats_out=...): this launch cannot produce column statistics (GEGLU / fp32 output / strided output)")
            self.colstats.pop(key, None)          # the buffer is being rewritten without statistics
        ln_rowsum = None
        if ln_gamma is not None:
            assert Recorder.gemm_ln_supported(M, N, kdim, geglu, self.big_min) and conv is None and a1 is None and not colstats and splitk == 1
            ln_rowsum = w.float().sum(1).contiguous()            # of the fp16 values the MFMAs see (w already carries gamma: fold_layernorm)
        if a_norm is not None:
            assert conv is not None and a_norm.dtype == torch.float32 and a_norm.is_contiguous() and a_norm.shape == (geo[0], 2, c0 + c1), a_norm.shape
            assert Recorder.gn_conv_supported(geo, M, N, c0, c1, big_min, splitk), "gemm(a_norm=...) needs the LDS-resident-patch conv (gn_conv_supported)"
        p = GemmParams(_ptr(a), _ptr(a1), c0, c1, lda0, lda1, _ptr(w), _ptr(bias), _ptr(rowadd), rowadd_ld, _ptr(residual), ldr,
                       _ptr(out), ldc, M, N, taps, *geo, act, int(out_f32), int(geglu), splitk, _ptr(ws), _ptr(cs), _ptr(ln_rowsum), float(ln_eps), big_min if big_min > 0 else -1,
                       _ptr(a_norm), int(a_norm_act))
        self.keep.extend(t for t in (a, a1, w, bias, rowadd, residual, out, ln_rowsum, a_norm) if t is not None)
        # the symbol rocprofv3 shows for this launch and its workgroup count (4th tag field: bench.py separates the chip-filling launches of the
        # one-per-CU tile from the half-chip ones): asked from the library's own dispatch code, not restated here
        try:
            name, wgs = _kernel_info(self.lib.pv_gemm_conv_kernel_info, p)
        except ValueError:              # a block the library rejects: recorded all the same - run() raises HipLaunchError, as for every other entry point
            name, wgs = "pv_gemm_conv", 0
        self._add(self.lib.pv_gemm_conv, p, tag=(name, 2.0 * M * N * kdim, 2.0 * (M * (c0 + c1) + N * kdim + M * n_out), wgs))
        return out

    #: LayerNorm folded into the 256-row-tile Linear launches (norm1 -> qkv, norm3 -> GEGLU at the 32 x 32 / 16 x 16 levels).  OFF by default: it removes
    #: 28 LayerNorm launches and 1.2 GB of HBM traffic per step and is exact to the same 2.5e-4, but measured 31.05 -> 30.98 steps/s same box (the
    #: v_dot2 row sums ride in the MFMA segments of a pair whose partner is issuing its LDS-DMA: not free).  PV_GEMM_LN=1 enables it.
    GEMM_LN = os.environ.get("PV_GEMM_LN", "0") != "0"

    @staticmethod
    def _probe_gemm(**fields) -> str:
        """Symbol pv_gemm_conv would launch for a parameter block with these fields (pointers are placeholders: the query never touches them), '' if the
        library rejects the block."""
        p = GemmParams()
        for k in ("a0", "w", "out"):
            setattr(p, k, 0x1000)
        for k, v in fields.items():
            setattr(p, k, v)
        try:
            return _kernel_info(load().pv_gemm_conv_kernel_info, p)[0]
        except ValueError:
            return ""

    @staticmethod
    def gemm_ln_supported(M: int, N: int, K: int, geglu: bool, big_min: Optional[int] = None) -> bool:
        """Where ``gemm(ln_gamma=...)`` exists: the launches the library puts on the LayerNorm instantiation of the 256-row tile's Linear modes (asked
        from pv_gemm_conv_kernel_info: a block carrying ``ln_rowsum`` that lands on ``big_tile_kernel<..., true>``)."""
        big_min = 0 if big_min is None else (big_min if big_min > 0 else -1)
        name = Recorder._probe_gemm(c0=K, lda0=K, N=N, ldc=N // 2 if geglu else N, M=M, taps=1, batch=1, hin=1, win=1, hout=M, wout=1, stride=1, pad=1,
                                    geglu=int(geglu), ln_rowsum=0x1000, ln_eps=1e-5, big_tile_min=big_min)
        return bool(Recorder.GEMM_LN and name.startswith("big_tile_kernel") and name.endswith("true>"))

    #: GroupNorm + SiLU folded into the 3x3 conv behind it (pv_gemm_params.a_norm, round 5): exact (bit-identical to GroupNorm-apply + conv), removes the
    #: 64 x 64 level's 20 GroupNorm-apply launches per step - and measures 32.3 -> 32.1 steps/s same box (three rounds): the conv pays ~600 cycles per
    #: normalised patch piece (+15 % per conv: 116.5 vs 101.5 us), as much as the launches it removes (profiles/r05_gnfold_*.txt).  OFF by default;
    #: PV_GN_FOLD=1 enables it; the entry point stays tested.
    GN_FOLD = os.environ.get("PV_GN_FOLD", "0") != "0"

    @staticmethod
    def gn_conv_supported(geo, M: int, N: int, c0: int, c1: int, big_min: int, splitk: int = 1) -> bool:
        """Where ``gemm(a_norm=...)`` exists: the launches the library puts on the GroupNorm-folding modes (5 / 6) of the LDS-resident-patch conv (asked from
        pv_gemm_conv_kernel_info with a block that carries ``a_norm``; the library rejects such a block everywhere else)."""
        if not Recorder.GN_FOLD:
            return False
        name = Recorder._probe_gemm(a1=0x1000 if c1 else 0, c0=c0, c1=c1, lda0=c0, lda1=c1, N=N, ldc=N, M=M, taps=9, batch=geo[0], hin=geo[1], win=geo[2],
                                    hout=geo[3], wout=geo[4], stride=geo[5], upsample=geo[6], pad=geo[7], splitk=splitk, splitk_ws=0x1000 if splitk > 1 else 0,
                                    big_tile_min=big_min if big_min > 0 else -1, a_norm=0x1000, a_norm_act=ACT_SILU)
        return name.startswith("big_tile_kernel") and (", 8, 5, " in name or ", 8, 6, " in name)

    def groupnorm_table(self, x: torch.Tensor, gamma, beta, *, batch: int, hw: int, x1: Optional[torch.Tensor] = None, eps=1e-5, groups=32) -> Optional[torch.Tensor]:
        """The GroupNorm of ``x`` (| ``x1``) as a per-(image, channel) scale / shift table, fp32 [batch][2][C], for ``gemm(a_norm=...)``; None when the inputs
        carry no column statistics (the caller then runs ``groupnorm``)."""
        ld0, c0 = _rows(x)
        ld1, c1 = _rows(x1) if x1 is not None else (0, 0)
        cs0 = self.colstats.get((x.data_ptr(), batch * hw, c0)) if ld0 == c0 else None
        cs1 = self.colstats.get((x1.data_ptr(), batch * hw, c1)) if (x1 is not None and ld1 == c1) else None
        if not (hw % 64 == 0 and cs0 is not None and (x1 is None or cs1 is not None)):
            return None
        partial = self.empty((batch, 1, groups, 2), torch.float32)
        table = self.empty((batch, 2, c0 + c1), torch.float32)
        p = GroupNormParams(_ptr(x), _ptr(x1), c0, c1, ld0, ld1, batch, hw, groups, 1, _ptr(partial), _ptr(gamma), _ptr(beta),
                            float(eps), ACT_NONE, None, _ptr(cs0), _ptr(cs1))
        self.keep.extend(t for t in (x, x1, gamma, beta) if t is not None)
        self._add(self.lib.pv_groupnorm_scale_shift, p, _ptr(table))
        return table

    @staticmethod
    def fold_layernorm(w: torch.Tensor, bias: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor):
        """LayerNorm's affine part folded into the Linear behind it: (w diag(gamma), bias + w . beta), w fp16 [N][K] in NATURAL row order."""
        w32 = w.float()
        b = w32 @ beta.float().to(w.device)
        if bias is not None:
            b = b + bias.float()
        return (w32 * gamma.float().to(w.device)[None, :]).to(torch.float16).contiguous(), b.contiguous()

    def groupnorm(self, x: torch.Tensor, gamma, beta, *, batch: int, hw: int, x1: Optional[torch.Tensor] = None, eps=1e-5,
                  act=ACT_NONE, groups=32, return_stats=False) -> torch.Tensor:
        """``return_stats``: also return the statistics buffer ((mean, rstd) per (image, group) at ``stats[b, g*2 : g*2+2]``) for
        ``groupnorm_backward``."""
        ld0, c0 = _rows(x)
        ld1, c1 = _rows(x1) if x1 is not None else (0, 0)
        splits = 64
        while splits > 1 and (hw % splits or hw // splits < 16):
            splits //= 2
        partial = self.empty((batch, splits, groups, 2), torch.float32)
        y = self.empty((batch * hw, c0 + c1), torch.float16)
        cs0 = self.colstats.get((x.data_ptr(), batch * hw, c0)) if ld0 == c0 else None
        cs1 = self.colstats.get((x1.data_ptr(), batch * hw, c1)) if (x1 is not None and ld1 == c1) else None
        from_cs = hw % 64 == 0 and cs0 is not None and (x1 is None or cs1 is not None)
        p = GroupNormParams(_ptr(x), _ptr(x1), c0, c1, ld0, ld1, batch, hw, groups, splits, _ptr(partial), _ptr(gamma), _ptr(beta),
                            float(eps), act, _ptr(y), _ptr(cs0) if from_cs else None, _ptr(cs1) if from_cs else None)
        self.keep.extend(t for t in (x, x1, gamma, beta) if t is not None)
        self._add(self.lib.pv_groupnorm_stats_from_colstats if from_cs else self.lib.pv_groupnorm_stats, p)
        self._add(self.lib.pv_groupnorm_apply, p)
        return (y, partial.view(batch, -1)) if return_stats else y

    def layernorm(self, x: torch.Tensor, gamma, beta, *, eps=1e-5, act=ACT_NONE, out=None) -> torch.Tensor:
        ldx, cols = _rows(x)
        if out is None:
            out = self.empty((x.shape[0], cols), torch.float16)
        ldy, _ = _rows(out)
        p = LayerNormParams(_ptr(x), ldx, _ptr(out), ldy, _ptr(gamma), _ptr(beta), x.shape[0], cols, float(eps), act)
        self.keep.extend((x, gamma, beta, out))
        self._add(self.lib.pv_layernorm, p)
        return out

    def attention(self, q, k, v, *, batch, heads, nq, nk, d, causal=False, out=None, lse=None) -> torch.Tensor:
        """``lse``: optional fp32 [batch, heads, nq] buffer that receives the log-sum-exp ``attention_backward`` needs."""
        ldq, _ = _rows(q)
        ldk, _ = _rows(k)
        ldv, _ = _rows(v)
        if out is None:
            out = self.empty((batch * nq, heads * d), torch.float16)
        ldo, _ = _rows(out)
        p = AttnParams(_ptr(q), _ptr(k), _ptr(v), ldq, ldk, ldv, _ptr(out), ldo, batch, heads, nq, nk, d, int(causal), _ptr(lse))
        self.keep.extend(t for t in (q, k, v, out, lse) if t is not None)
        # the symbol rocprofv3 shows for this launch and its workgroup count: pv_attn.hip's own rule (choose_attn), asked from the library
        try:
            name, wgs = _kernel_info(self.lib.pv_attention_kernel_info, p)
        except ValueError:              # rejected block: run() raises HipLaunchError
            name, wgs = "pv_attention", 0
        self._add(self.lib.pv_attention, p, tag=(name, 4.0 * batch * heads * nq * nk * d * (0.5 if causal else 1.0), 2.0 * batch * heads * d * (2 * nq + 2 * nk), wgs))
        return out

    # ---- backward of the stock blocks the training gradient crosses (pv_train.hip) ----
    def attention_backward(self, q, k, v, out, dout, lse, *, batch, heads, nq, nk, d, causal=False, dq=None, dk=None, dv=None):
        """(dq, dk, dv) fp16 rows of softmax(q k^T / sqrt(d)) v; pass column slices of one buffer to get [dq | dk | dv]."""
        C_ = heads * d
        dq = self.empty((batch * nq, C_)) if dq is None else dq
        dk = self.empty((batch * nk, C_)) if dk is None else dk
        dv = self.empty((batch * nk, C_)) if dv is None else dv
        delta = self.empty((batch, heads, nq), torch.float32)
        qs = self.empty((batch * nq, C_))
        # workspace of the 8-wave staggered passes (pv_attnbwd.hip, d = 40 / 80): scaled queries and dO head-major in 48- / 112-column rows
        ws_cols, own = {40: (48, 512), 80: (112, 256)}.get(d, (0, 1))
        # ONE scratch buffer per Recorder, shared by every attention_backward call of the plan: it is written and consumed inside one stream-ordered call
        # (the plan's calls run in order on one stream), so 0.8 GB of per-layer buffers at bs = 16 become the largest layer's ~100 MB.  A later call that needs
        # more gets a new, larger buffer (earlier calls keep theirs); a backward plan meets its largest layer first.
        ws = None
        if ws_cols and not causal and nq % own == 0 and nk % own == 0:
            need = 2 * batch * heads * nq * ws_cols
            if getattr(self, "_attn_bwd_ws", None) is None or self._attn_bwd_ws.numel() < need:
                self._attn_bwd_ws = self.empty((need,))
            ws = self._attn_bwd_ws[:need]
        p = AttnBwdParams(_ptr(q), _ptr(k), _ptr(v), _rows(q)[0], _rows(k)[0], _rows(v)[0], _ptr(out), _rows(out)[0], _ptr(dout), _rows(dout)[0],
                          _ptr(lse), _ptr(delta), _ptr(qs), C_, _ptr(dq), _ptr(dk), _ptr(dv), _rows(dq)[0], _rows(dk)[0], _rows(dv)[0], batch, heads, nq, nk, d,
                          int(causal), _ptr(ws), 0 if ws is None else ws.numel() * 2)
        self.keep.extend((q, k, v, out, dout, lse, dq, dk, dv))
        # five matrix products of 2 nq nk d each (S, dP, dV, dK, dQ): the usual algorithmic count of a flash backward
        self._add(self.lib.pv_attention_backward, p, tag=("pv_attention_backward", 10.0 * batch * heads * nq * nk * d * (0.5 if causal else 1.0), 2.0 * batch * heads * d * (4 * nq + 4 * nk)))
        return dq, dk, dv

    def groupnorm_backward(self, x, dy, stats, gamma, beta, *, batch, hw, x1=None, act=ACT_NONE, groups=32, add0=None, add1=None,
                           want0=True, want1=True):
        """Data gradient of ``groupnorm`` (``stats`` = the ``stats`` it returned): (dx0, dx1) fp16, each + its ``add`` when given."""
        ld0, c0 = _rows(x)
        ld1, c1 = _rows(x1) if x1 is not None else (0, 0)
        Cc = c0 + c1
        splits = 64
        while splits > 1 and (hw % splits or hw // splits < 16):
            splits //= 2
        partial = self.empty((batch, splits, 2, Cc), torch.float32)
        sums = self.empty((batch, groups, 2), torch.float32)
        dx0 = self.empty((batch * hw, c0)) if want0 else None
        dx1 = self.empty((batch * hw, c1)) if (want1 and x1 is not None) else None
        p = GroupNormBwdParams(_ptr(x), _ptr(x1), c0, c1, ld0, ld1, batch, hw, groups, splits, _ptr(stats), stats.stride(0), _ptr(gamma), _ptr(beta),
                               act, _ptr(dy), _rows(dy)[0], _ptr(partial), _ptr(sums),
                               _ptr(dx0), c0, _ptr(add0), _rows(add0)[0] if add0 is not None else 0,
                               _ptr(dx1), c1, _ptr(add1), _rows(add1)[0] if add1 is not None else 0)
        self.keep.extend(t for t in (x, x1, dy, stats, gamma, beta, add0, add1) if t is not None)
        self._add(self.lib.pv_groupnorm_backward, p)
        return dx0, dx1

    def geglu_backward(self, h, dy):
        rows, n2 = h.shape
        dh = self.empty((rows, n2))
        self.keep.extend((h, dy))
        self._add(self.lib.pv_geglu_backward, _ptr(h), _rows(h)[0], _ptr(dy), _rows(dy)[0], _ptr(dh), n2, rows, n2 // 2)
        return dh

    def act_backward(self, x, dy, act):
        rows, cols = x.shape
        dx = self.empty((rows, cols))
        self.keep.extend((x, dy))
        self._add(self.lib.pv_act_backward, _ptr(x), _rows(x)[0], _ptr(dy), _rows(dy)[0], _ptr(dx), cols, rows, cols, act)
        return dx

    def act_forward(self, x, act):
        rows, cols = x.shape
        y = self.empty((rows, cols))
        self.keep.append(x)
        self._add(self.lib.pv_act_forward, _ptr(x), _rows(x)[0], _ptr(y), cols, rows, cols, act)
        return y

    def dropout(self, x, *, p, rng, site, copies=1, backward=False, add=None):
        """Forward: [rows, copies * cols] independently masked, 1/(1-p)-scaled copies of x; backward: x = the gradient of that, returns its
        masked sum over the copies (+ add).  Same (rng, site) -> same masks."""
        rows = x.shape[0]
        cols = x.shape[1] // copies if backward else x.shape[1]
        out = self.empty((rows, cols if backward else cols * copies))
        self.keep.extend(t for t in (x, rng, add) if t is not None)
        self._add(self.lib.pv_dropout_f16, _ptr(x), _rows(x)[0], _ptr(out), _rows(out)[0], _ptr(add), _rows(add)[0] if add is not None else 0, rows, cols,
                  copies, float(p), _ptr(rng), int(site), int(backward))
        return out

    # ---- ArcFace identity loss / VAE-decoder backward pieces ----
    def col_affine(self, x, scale, shift=None):
        rows, cols = x.shape
        y = self.empty((rows, cols))
        self.keep.extend(t for t in (x, scale, shift) if t is not None)
        self._add(self.lib.pv_col_affine_f16, _ptr(x), _rows(x)[0], _ptr(scale), _ptr(shift), _ptr(y), cols, rows, cols)
        return y

    def prelu(self, x, slope, dy=None):
        """dy None: prelu(x); else dy * prelu'(x).  ``slope``: fp32 device scalar (nn.PReLU().weight)."""
        rows, cols = x.shape
        out = self.empty((rows, cols))
        self.keep.extend(t for t in (x, slope, dy) if t is not None)
        self._add(self.lib.pv_prelu_f16, _ptr(x), _rows(x)[0], _ptr(dy), _rows(dy)[0] if dy is not None else 0, _ptr(slope), _ptr(out), cols, rows, cols)
        return out

    def maxpool2x2(self, x, *, batch, h, w, dy=None):
        c = x.shape[1]
        assert x.is_contiguous() and (dy is None or dy.is_contiguous())
        out = self.empty((batch * h * w, c)) if dy is not None else self.empty((batch * (h // 2) * (w // 2), c))
        self.keep.extend(t for t in (x, dy) if t is not None)
        self._add(self.lib.pv_maxpool2x2, _ptr(x), _ptr(dy), _ptr(out), batch, h, w, c)
        return out

    def gray_resize(self, x, *, size, mul=1.0, add=0.0):
        """fp32 (B, 3, H, W) (images may be strided) -> fp32 (B, 1, size, size): grayscale + bilinear resize (+ affine)."""
        B, _, H, W = x.shape
        assert x.dtype == torch.float32 and x.stride(3) == 1 and x.stride(2) == W and x.stride(1) == H * W
        y = self.empty((B, 1, size, size), torch.float32)
        self.keep.append(x)
        self._add(self.lib.pv_gray_resize, _ptr(x), x.stride(0), _ptr(y), B, H, W, size, float(mul), float(add))
        return y

    def gray_resize_backward(self, dy, *, h, w, mul=1.0):
        B, _, S, _ = dy.shape
        assert dy.dtype == torch.float32 and dy.stride(3) == 1 and dy.stride(2) == S
        dx = self.empty((B, 3, h, w), torch.float32)
        self.keep.append(dy)
        self._add(self.lib.pv_gray_resize_backward, _ptr(dy), dy.stride(0), _ptr(dx), B, h, w, S, float(mul))
        return dx

    def cosine_embedding_loss(self, e1, e2, *, target=1.0, gscale=1.0, want_grad=True):
        """(per-sample losses fp32 [B], de2 fp16 [B, dim] = gscale * d(mean loss)/d(e2))."""
        B, dim = e2.shape
        ls = self.empty((B,), torch.float32)
        de2 = self.empty((B, dim)) if want_grad else None
        assert e1.is_contiguous() and e2.is_contiguous() and e1.dtype == torch.float16 and e2.dtype == torch.float16
        self.keep.extend((e1, e2))
        self._add(self.lib.pv_cosine_embedding_loss, _ptr(e1), _ptr(e2), B, dim, float(target), float(gscale), _ptr(ls), _ptr(de2))
        return ls, de2

    def softmax_rows_backward(self, p, dp, *, scale):
        self.keep.extend((p, dp))
        self._add(self.lib.pv_softmax_rows_backward, _ptr(p), _rows(p)[0], _ptr(dp), _rows(dp)[0], p.shape[0], p.shape[1], float(scale))
        return dp

    def clamp_mask(self, y, dy, lo, hi):
        out = self.empty(tuple(y.shape), torch.float32)
        assert y.is_contiguous() and dy.is_contiguous()
        self.keep.extend((y, dy))
        self._add(self.lib.pv_clamp_mask_f32, _ptr(y), _ptr(dy), float(lo), float(hi), _ptr(out), y.numel())
        return out

    def add_rows(self, a, b, out=None):
        rows, cols = a.shape
        if out is None:
            out = self.empty((rows, cols))
        self.keep.extend((a, b, out))
        self._add(self.lib.pv_add_rows_f16, _ptr(a), _rows(a)[0], _ptr(b), _rows(b)[0], _ptr(out), _rows(out)[0], rows, cols)
        return out

    def dilate2x(self, x, *, batch, h, w):
        c = x.shape[1]
        z = self.empty((batch * 4 * h * w, c))
        self.keep.append(x)
        self._add(self.lib.pv_dilate2x, _ptr(x), _rows(x)[0], _ptr(z), batch, h, w, c)
        return z

    def pool2x_sum(self, g, *, batch, h, w, add=None):
        c = g.shape[1]
        out = self.empty((batch * h * w, c))
        self.keep.extend(t for t in (g, add) if t is not None)
        self._add(self.lib.pv_pool2x_sum, _ptr(g), _ptr(add), _rows(add)[0] if add is not None else 0, _ptr(out), batch, h, w, c)
        return out

    def sign(self, x, coef):
        out = self.empty(tuple(x.shape), torch.float32)
        self.keep.append(x)
        self._add(self.lib.pv_sign_f32, _ptr(x), float(coef), _ptr(out), x.numel())
        return out

    def gather_rows(self, x, idx, *, batch, seq, n_e, scale=1.0):
        dim = x.shape[1]
        out = self.empty((batch, n_e, dim), torch.float32)
        self.keep.extend((x, idx))
        self._add(self.lib.pv_gather_rows_f32, _ptr(x), _rows(x)[0], _ptr(idx), _ptr(out), batch, seq, n_e, dim, float(scale))
        return out

    def cross_attention(self, q, kt, vt, kip, vip, *, batch, heads, nq, nt, nip, d, w_text=1.0, w_ip=1.0, vnorm=None, out=None, fusion=None):
        if out is None:
            out = self.empty((batch * nq, heads * d), torch.float16)
        p = XAttnParams(_ptr(q), _rows(q)[0], _ptr(kt), _ptr(vt), _rows(kt)[0], _rows(vt)[0], _ptr(kip), _ptr(vip), _rows(kip)[0],
                        _rows(vip)[0], _ptr(out), _rows(out)[0], _ptr(vnorm), batch, heads, nq, nt, nip, d, float(w_text), float(w_ip),
                        _ptr(fusion))
        self.keep.extend(t for t in (q, kt, vt, kip, vip, out, vnorm, fusion) if t is not None)
        self._add(self.lib.pv_cross_attention, p, tag=("pv_cross_attention", 4.0 * batch * nq * (nt + nip) * heads * d, 2.0 * 2 * batch * nq * heads * d))
        return out, p

    # ---- attn2 branch of the C = 1280 / d = 160 layers: norm2 -> to_q -> dual-branch SDPA as ONE head-parallel launch (pv_xq.hip) ----
    XLNQ = os.environ.get("PV_XLNQ", "1") != "0"          # A/B switch

    @staticmethod
    def xattn_lnq_supported(C: int, heads: int, nt: int, nip: int) -> bool:
        """d = 160 (C = 1280: one head per 160-feature block) or d = 80 (C = 640: two heads per block)."""
        return Recorder.XLNQ and C % heads == 0 and C // heads in (160, 80) and 0 < nt <= 80 and 0 <= nip <= 16

    def cross_attention_lnq(self, hs, wq, kt, vt, kip, vip, *, batch, heads, nq, nt, nip, d=None, ln_gamma=None, ln_beta=None, ln_eps=1e-5,
                            w_text=1.0, w_ip=1.0, vnorm=None, fusion=None, out=None):
        """``ctx = SDPA_dual(to_q(LayerNorm(hs)), K, V)`` per head; ``wq``: to_q.weight fp16 [C][C].  norm2 is folded at plan-build time so
        that the kernel's GEMM reads the raw rows: gamma scales the columns of wq, ``q_bias = wq . beta``, and the kernel corrects with the
        row statistics it accumulates itself: ``rstd * (wq' . x - mean * rowsum(wq')) + q_bias``."""
        C = hs.shape[1]
        d = C // heads if d is None else d
        assert heads * d == C and d in (160, 80) and wq.shape == (C, C) and wq.dtype == torch.float16
        if out is None:
            out = self.empty((batch * nq, C), torch.float16)
        q_bias = rowsum = None
        if ln_gamma is not None:
            w32 = wq.float()
            q_bias = (w32 @ ln_beta.float().to(wq.device)).contiguous()
            wq = (w32 * ln_gamma.float().to(wq.device)[None, :]).to(torch.float16)
            rowsum = wq.float().sum(1).contiguous()          # of the fp16 values the MFMAs see
        wq = wq.contiguous()
        p = XAttnLnqParams(_ptr(hs), _rows(hs)[0], int(ln_gamma is not None), float(ln_eps), _ptr(wq), _ptr(q_bias), _ptr(rowsum), _ptr(kt), _ptr(vt),
                           _rows(kt)[0], _rows(vt)[0], _ptr(kip), _ptr(vip), _rows(kip)[0] if kip is not None else 0, _rows(vip)[0] if vip is not None else 0,
                           _ptr(out), _rows(out)[0], _ptr(vnorm), batch, nq, heads, d, nt, nip, float(w_text), float(w_ip), _ptr(fusion))
        self.keep.extend(t for t in (hs, wq, q_bias, rowsum, kt, vt, kip, vip, out, vnorm, fusion) if t is not None)
        M = batch * nq
        self._add(self.lib.pv_cross_attention_lnq, p, tag=("xattn_lnq_kernel", 2.0 * M * C * C + 4.0 * M * (nt + nip) * C, 2.0 * (2 * M * C + C * C)))
        return out, p

    # ---- fused attn2 branch (norm2 -> to_q -> dual-branch SDPA -> to_out + residual): C = 320 / d = 40 and C = 640 / d = 80 layers ----
    XFUSED_WIDTHS = tuple(int(w) for w in os.environ.get("PV_XFUSED_WIDTHS", "320,640").split(",") if w)   # A/B switch

    @staticmethod
    def xattn_fused_supported(C: int, heads: int, nq: int, nt: int, nip: int) -> bool:
        return C in Recorder.XFUSED_WIDTHS and heads == 8 and nq % 128 == 0 and 64 < nt <= 80 and 0 < nip <= 16

    def pack_wo_for_fused(self, wo: torch.Tensor) -> torch.Tensor:
        """to_out[0].weight [C][C] with its columns in the fused kernel's context-slot order (pv_xattn_fused_wo_slot)."""
        C = wo.shape[1]
        idx = torch.tensor([self.lib.pv_xattn_fused_wo_slot(s) for s in range(C)], device=wo.device)
        return wo[:, idx].contiguous()

    def xattn_pack_kv(self, kt, vt, kip, vip, *, batch, heads, d, nt, nip, vnorm=None):
        """K / V images of the fused kernel (once per conditioning) + to_v_ip_norm."""
        C = heads * d
        kimg = self.empty((batch * heads * 96 * (64 if d == 40 else 128),), torch.float16)   # 64 / 128 contraction slots per key row
        vimg = self.empty((batch * (C // 80) * 96 * 80,), torch.float16)
        self.keep.extend(t for t in (kt, vt, kip, vip, vnorm) if t is not None)
        self._add(self.lib.pv_xattn_pack_kv, _ptr(kt), _ptr(vt), _rows(kt)[0], _rows(vt)[0], _ptr(kip), _ptr(vip), _rows(kip)[0],
                  _rows(vip)[0], _ptr(kimg), _ptr(vimg), _ptr(vnorm), batch, heads, d, nt, nip)
        return kimg, vimg

    def cross_attention_fused(self, hs, wq, wo_packed, bias_o, kimg, vimg, *, batch, nq, heads, d, nt, nip, ln_gamma=None, ln_beta=None,
                              ln_eps=1e-5, w_text=1.0, w_ip=1.0, fusion=None, out=None):
        """``wq``: to_q.weight fp16 [C][C].  With ``ln_gamma`` / ``ln_beta`` the kernel applies norm2 in front of to_q; its affine part
        is folded here, once at plan-build time: gamma scales the columns of wq, beta becomes a bias on q (= wq . beta)."""
        C = heads * d
        if out is None:
            out = self.empty((batch * nq, C), torch.float16)
        q_bias = None
        if ln_gamma is not None:
            w32 = wq.float()
            q_bias = (w32 @ ln_beta.float()).contiguous()
            wq = (w32 * ln_gamma.float()[None, :]).to(torch.float16).contiguous()
        p = XAttnFusedParams(_ptr(hs), _rows(hs)[0], int(ln_gamma is not None), float(ln_eps), _ptr(wq), _ptr(q_bias), _ptr(wo_packed), _ptr(bias_o),
                             _ptr(kimg), _ptr(vimg), _ptr(out), _rows(out)[0], batch, nq, heads, d, nt, nip, float(w_text), float(w_ip),
                             _ptr(fusion), 128 if (C == 640 and 0 < self.big_min <= 128) else 0)   # half-chip launches when the plan runs beside its CFG twin
        # (also where 128-row workgroups do NOT fill that half: configs[4]'s per-rank shape has 72 of them per launch and is still 0.7 % of a step faster
        #  with them than with 144 64-row workgroups, which stream the weights twice per row: profiles/r06_loop_ab_cfg4_splitk.txt)
        self.keep.extend(t for t in (hs, wq, q_bias, wo_packed, bias_o, kimg, vimg, fusion, out) if t is not None)
        M = batch * nq
        flops = 4.0 * M * C * C + 4.0 * M * (nt + nip) * C           # to_q + to_out + both SDPA products (dense-counted)
        self._add(self.lib.pv_cross_attention_fused, p, tag=("xattn_fused_kernel<%d, %s>" % (C, "true" if nip == 1 else "false"), flops, 2.0 * (3 * M * C + 2 * C * C)))
        return out, p

    # ---- LayerNorm + Linear (+ GEGLU) of the K = 320 transformer layers as one row-owning launch (pv_rowgemm.hip) ----
    @staticmethod
    def row_gemm_supported(K: int, N: int) -> bool:
        return K == 320 and N % 320 == 0

    def row_gemm(self, x, w, *, bias=None, ln_gamma=None, ln_beta=None, ln_eps=1e-5, geglu=False, out=None, x_norm=None, rows_per_image=0):
        """``out = epi(LayerNorm(x) . w^T + bias)``: ``w`` fp16 [N][320] (for ``geglu``: rows and bias already in ``pack_geglu_rows`` order).
        With ``ln_gamma`` / ``ln_beta`` the kernel normalises the rows in registers; the affine part is folded here, once at plan-build
        time: gamma scales the columns of w, w . beta joins the bias.  ``x_norm`` (``groupnorm_table``'s result) + ``rows_per_image``: the rows are
        GroupNorm-ed in registers instead (Transformer2DModel.norm -> proj_in on the raw block output)."""
        M, K = x.shape
        if x_norm is not None:
            assert ln_gamma is None and rows_per_image % 128 == 0 and M % rows_per_image == 0 and x_norm.shape == (M // rows_per_image, 2, K) and x_norm.dtype == torch.float32
        N = w.shape[0]
        assert Recorder.row_gemm_supported(K, N) and w.shape[1] == K and w.dtype == torch.float16
        ln = ln_gamma is not None
        b32 = None if bias is None else bias.detach().float()
        if ln:
            wf = w.float()
            fold = wf @ ln_beta.detach().float().to(w.device)
            b32 = fold if b32 is None else b32 + fold
            w = (wf * ln_gamma.detach().float().to(w.device)[None, :]).to(torch.float16)
        w = w.contiguous()
        b32 = None if b32 is None else b32.contiguous()
        n_out = N // 2 if geglu else N
        if out is None:
            out = self.empty((M, n_out), torch.float16)
        p = _lib.RowGemmParams(_ptr(x), _rows(x)[0], M, K, N, _ptr(w), _ptr(b32), 1 if ln else 0, float(ln_eps), 1 if geglu else 0, _ptr(out), _rows(out)[0],
                               _ptr(x_norm), int(rows_per_image))
        self.keep.extend(t for t in (x, w, b32, out, x_norm) if t is not None)
        self._add(self.lib.pv_row_gemm, p, tag=("row_gemm_kernel<%s>" % ("true" if geglu else "false"), 2.0 * M * N * K, 2.0 * (M * K + N * K + M * n_out)))
        return out

    # ---- backward of PhotoVerse's own trainable modules (pv_backward.hip) ----
    def cross_attention_backward(self, q, kt, vt, kip, vip, dout, *, batch, heads, nq, nt, nip, d, w_text=1.0, w_ip=1.0, fusion=None,
                                 out_scale=1.0, vnorm_coef=0.0, vnorm_grad=None):
        """Gradients of the dual-branch SDPA: dq fp16 [B*nq, C]; [dK_t | dV_t] fp32 [B*nt, 2C]; [dK_ip | dV_ip] fp32 [B*nip, 2C]
        (returned as dq, dkv_text, dkv_ip)."""
        C_ = heads * d
        dq = self.empty((batch * nq, C_), torch.float16)
        dkv_t = self.empty((batch * nt, 2 * C_), torch.float32)
        dkv_i = self.empty((batch * nip, 2 * C_), torch.float32)
        dkt, dvt, dkip, dvip = dkv_t[:, :C_], dkv_t[:, C_:], dkv_i[:, :C_], dkv_i[:, C_:]
        nchunk = (nq + 511) // 512
        partial = self.empty((batch * heads * nchunk * 2 * 96 * d,), torch.float32)
        stats = self.empty((batch, heads, nq, 4), torch.float32)
        p = XAttnBwdParams(_ptr(q), _rows(q)[0], _ptr(kt), _ptr(vt), _rows(kt)[0], _rows(vt)[0], _ptr(kip), _ptr(vip), _rows(kip)[0], _rows(vip)[0],
                           _ptr(dout), _rows(dout)[0], _ptr(dq), C_, _ptr(partial), _ptr(stats), _ptr(dkt), _ptr(dvt), _ptr(dkip), _ptr(dvip), 2 * C_, 2 * C_, batch, heads, nq, nt,
                           nip, d, float(w_text), float(w_ip), _ptr(fusion), float(out_scale), float(vnorm_coef), _ptr(vnorm_grad))
        self.keep.extend(t for t in (q, kt, vt, kip, vip, dout, fusion, vnorm_grad) if t is not None)
        self._add(self.lib.pv_cross_attention_backward, p, tag=("pv_cross_attention_backward", 10.0 * batch * nq * (nt + nip) * heads * d, 2.0 * 3 * batch * nq * heads * d))
        return dq, dkv_t, dkv_i

    def transpose(self, x, rows_pad=None):
        """fp16 [R, C] -> [C, rows_pad] (zero padded rows): the operand layout of the weight-gradient GEMMs."""
        ldx, cols = _rows(x)
        rows = x.shape[0]
        rp = rows_pad or (rows + 63) // 64 * 64
        out = self.empty((cols, rp), torch.float16)
        self.keep.append(x)
        self._add(self.lib.pv_transpose_f16, _ptr(x), ldx, rows, cols, _ptr(out), rp, rp)
        return out

    def wgrad(self, dy16, x16, scale=None):
        """dW [N, K] fp32 = dy16[M, N]^T . x16[M, K]: ``pv_wgrad_tn`` (MFMA, operands read transposed from LDS - no transposed copies) over
        row slabs + the fixed-order slab sum (which also applies ``scale``).  N, K multiples of 8."""
        (lddy, n), (ldx, k) = _rows(dy16), _rows(x16)
        m = dy16.shape[0]
        assert x16.shape[0] == m
        # the output is tiny and the contraction is the row count: split the rows over workgroups (a 320 x 320 gradient over 65536 rows is
        # 9 tiles) - about 512 workgroups, at least 8 steps of 64 rows each, slabs summed in order
        tiles = ((n + 127) // 128) * ((k + 127) // 128)
        steps = (m + 63) // 64
        nsplit = max(1, min(steps // 8, 512 // tiles, 128))
        rps = ((steps + nsplit - 1) // nsplit) * 64
        nsplit = (m + rps - 1) // rps
        out = self.empty((n, k), torch.float32)
        sc = 1.0 if scale is None else float(scale)          # applied by the slab sum (a one-slab sum when the rows were not split)
        part = out if (nsplit == 1 and sc == 1.0) else self.empty((nsplit, n, k), torch.float32)
        self.keep.extend((dy16, x16))
        self._add(self.lib.pv_wgrad_tn, _ptr(dy16), lddy, _ptr(x16), ldx, m, n, k, _ptr(part), nsplit, rps,
                  tag=("pv_wgrad_tn", 2.0 * m * n * k, 2.0 * m * (n + k) + 4.0 * nsplit * n * k))
        if part is not out:
            self._add(self.lib.pv_reduce_blocks, _ptr(part), nsplit, n * k, sc, _ptr(out))
        return out

    def layernorm_backward(self, x, dy, gamma, beta, *, eps=1e-5, act=ACT_NONE, want_affine=True, dy_group=1, dy_skip=0, dy_scale=1.0, add=None):
        """dx fp16 [rows, cols] and (dgamma, dbeta) fp32 [2, cols].  ``dy_group`` > 1: dy has rows / dy_group rows, row r uses
        dy[r // dy_group] * dy_scale, except the first ``dy_skip`` rows of each group (zero).  ``add``: fp16 [rows, cols] added to dx (the
        gradient x already holds from its other consumers)."""
        rows, cols = x.shape
        dx = self.empty((rows, cols), torch.float16)
        rpw = 2 if (want_affine and rows >= 2048) else 1   # rows per wave: halves the dgamma / dbeta partial blocks; 8 left 129 workgroups for 4112 rows (130 us)
        nblk = (rows + 4 * rpw - 1) // (4 * rpw)
        part = self.empty((nblk, 2, cols), torch.float32) if want_affine else None
        p = LayerNormBwdParams(_ptr(x), _rows(x)[0], _ptr(dy), _rows(dy)[0], _ptr(dx), cols, _ptr(gamma), _ptr(beta), _ptr(part), rows, cols,
                               float(eps), act, int(dy_group), int(dy_skip), float(dy_scale), rpw, _ptr(add), _rows(add)[0] if add is not None else 0)
        self.keep.extend(t for t in (x, dy, gamma, beta, add) if t is not None)
        self._add(self.lib.pv_layernorm_backward, p)
        dgb = None
        if want_affine:
            dgb = self.empty((2, cols), torch.float32)
            self._add(self.lib.pv_reduce_blocks, _ptr(part), nblk, 2 * cols, 1.0, _ptr(dgb))
        return dx, dgb

    def colsum(self, x16):
        """fp32 [C] column sums of fp16 rows (bias gradients), deterministic."""
        rows, cols = x16.shape
        nblk = max(1, min(256, rows // 64))
        part = self.empty((nblk, cols), torch.float32)
        out = self.empty((cols,), torch.float32)
        self.keep.append(x16)
        self._add(self.lib.pv_colsum_f16, _ptr(x16), _rows(x16)[0], rows, cols, _ptr(part), nblk, _ptr(out))
        return out

    def geglu(self, x, out=None):
        ldx, n2 = _rows(x)
        n = n2 // 2
        if out is None:
            out = self.empty((x.shape[0], n), torch.float16)
        self.keep.extend((x, out))
        self._add(self.lib.pv_geglu, _ptr(x), ldx, _ptr(out), _rows(out)[0], x.shape[0], n)
        return out

    def timestep_embedding(self, timesteps, state, rows, dim):
        out = self.empty((rows, dim), torch.float16)
        self.keep.extend(t for t in (timesteps, state) if t is not None)
        self._add(self.lib.pv_timestep_embedding, _ptr(timesteps), _ptr(state), rows, dim, _ptr(out))
        return out

    def softmax_rows(self, x, *, scale):
        ld, cols = _rows(x)
        self.keep.append(x)
        self._add(self.lib.pv_softmax_rows, _ptr(x), ld, x.shape[0], cols, float(scale))
        return x

    def pointwise_nchw(self, x, w, bias, *, batch, cin, cout, hw):
        out = self.empty((batch, cout, hw), torch.float32)
        self.keep.extend((x, w, bias))
        self._add(self.lib.pv_pointwise_nchw, _ptr(x), _ptr(w), _ptr(bias), _ptr(out), batch, cin, cout, hw)
        return out

    def affine_rows(self, x, ca, y=None, cb=None, out=None):
        """out[b] = ca[b] * x[b] (+ cb[b] * y[b]); x / y / out fp32 [B, ...] contiguous, ca / cb fp32 [B]."""
        if out is None:
            out = self.empty(tuple(x.shape), torch.float32)
        B = x.shape[0]
        self.keep.extend(t for t in (x, y, ca, cb, out) if t is not None)
        self._add(self.lib.pv_affine_rows_f32, _ptr(x), _ptr(y), _ptr(ca), _ptr(cb), _ptr(out), x.numel() // B, B)
        return out

    def posterior_sample(self, moments, eps, out=None):
        """moments fp32 [B, 2c, h, w] (mean | logvar), eps fp32 [B, c, h, w] -> mean + exp(0.5 clamp(logvar)) eps."""
        B, c2 = moments.shape[0], moments.shape[1]
        if out is None:
            out = self.empty(tuple(eps.shape), torch.float32)
        self.keep.extend((moments, eps, out))
        self._add(self.lib.pv_posterior_sample, _ptr(moments), _ptr(eps), _ptr(out), B, eps.numel() // B)
        return out

    def reduce_mean(self, a, b=None, *, mode="mean", out=None):
        """Deterministic mean(a) / mean|a| / mean((a-b)^2) -> fp32 scalar tensor on the device."""
        m = {"mean": 0, "abs": 1, "mse": 2}[mode]
        nb = max(1, min(1024, (a.numel() + 4095) // 4096))
        partial = self.empty((nb,), torch.float32)
        if out is None:
            out = self.empty((1,), torch.float32)
        assert a.is_contiguous() and (b is None or (b.is_contiguous() and b.dtype == a.dtype and b.numel() == a.numel()))
        self.keep.extend(t for t in (a, b, out) if t is not None)
        self._add(self.lib.pv_reduce_mean, _ptr(a), _ptr(b), m, int(a.dtype == torch.float16), a.numel(), _ptr(partial), nb, _ptr(out))
        return out

    def reduce_sumsq(self, a, *, out, scale=1.0):
        """out[0] = scale * sum(a^2) for an fp32 tensor, deterministic."""
        nb = max(1, min(1024, (a.numel() + 4095) // 4096))
        partial = self.empty((nb,), torch.float32)
        self.keep.extend((a, out))
        self._add(self.lib.pv_reduce_sumsq, _ptr(a), a.numel(), float(scale), _ptr(partial), nb, _ptr(out))
        return out

    def clamp_(self, x, lo, hi):
        self.keep.append(x)
        self._add(self.lib.pv_clamp_f32, _ptr(x), float(lo), float(hi), x.numel())
        return x

    def im2col3x3(self, x, *, batch, cin, h, wd, kpad):
        out = self.empty((batch * h * wd, kpad), torch.float16)
        self.keep.append(x)
        self._add(self.lib.pv_im2col3x3, _ptr(x), _ptr(out), batch, cin, h, wd, kpad)
        return out

    def conv_out(self, x, w, bias, *, batch, cin, h, wd, cout, out=None):
        if out is None:
            out = self.empty((batch, cout, h, wd), torch.float32)
        self.keep.extend((x, w, bias, out))
        self._add(self.lib.pv_conv_out, _ptr(x), _ptr(w), _ptr(bias), _ptr(out), batch, cin, h, wd, cout)
        return out

    def cast_to_f16(self, x, out=None):
        if out is None:
            out = self.empty(tuple(x.shape), torch.float16)
        self.keep.extend((x, out))
        self._add(self.lib.pv_cast_f32_to_f16, _ptr(x), _ptr(out), x.numel())
        return out

    def cast_to_f32(self, x, out=None):
        if out is None:
            out = self.empty(tuple(x.shape), torch.float32)
        self.keep.extend((x, out))
        self._add(self.lib.pv_cast_f16_to_f32, _ptr(x), _ptr(out), x.numel())
        return out

    def rows_mean(self, x, *, groups, count, group_rows=None, out=None, accumulate=False):
        ldx, cols = _rows(x)
        if out is None:
            out = self.empty((groups, cols), torch.float16)
        self.keep.extend((x, out))
        self._add(self.lib.pv_rows_mean, _ptr(x), ldx, _ptr(out), _rows(out)[0], groups, count, group_rows or count, cols, int(accumulate))
        return out

    def patchify(self, pixels, *, batch, ch, img, patch, kpad):
        out = self.empty((batch * (img // patch) ** 2, kpad), torch.float16)
        self.keep.append(pixels)
        self._add(self.lib.pv_patchify, _ptr(pixels), _ptr(out), batch, ch, img, patch, kpad)
        return out

    def clip_vision_embed(self, patches, cls, pos, *, batch, ntok, dim):
        out = self.empty((batch * ntok, dim), torch.float16)
        self.keep.extend((patches, cls, pos))
        self._add(self.lib.pv_clip_vision_embed, _ptr(patches), _ptr(cls), _ptr(pos), _ptr(out), batch, ntok, dim)
        return out

    def clip_text_embed(self, ids, tok, pos, concept, placeholder_idx, *, n_concept, batch, seq, dim):
        out = self.empty((batch * seq, dim), torch.float16)
        self.keep.extend(t for t in (ids, tok, pos, concept, placeholder_idx) if t is not None)
        self._add(self.lib.pv_clip_text_embed, _ptr(ids), _ptr(tok), _ptr(pos), _ptr(concept), _ptr(placeholder_idx), n_concept,
                  _ptr(out), batch, seq, dim)
        return out

    def cfg_dpm_step(self, eps_u, eps_c, latents, x0_prev, coef, state, guidance):
        self.keep.extend((eps_u, eps_c, latents, x0_prev, coef, state))
        self._add(self.lib.pv_cfg_dpm_step, _ptr(eps_u), _ptr(eps_c), _ptr(latents), _ptr(x0_prev), _ptr(coef), _ptr(state),
                  float(guidance), latents.numel())

    def fusion_draw(self, state, rng, forced, out, *, n_layers, rule1, rule2, scale, only_last_step):
        """Device-side grad-mode fusion draw (attention_processor.py:413-420 without the host sync): fills out[n_layers][2]."""
        self.keep.extend(t for t in (state, rng, forced, out) if t is not None)
        self._add(self.lib.pv_fusion_draw, _ptr(state), _ptr(rng), _ptr(forced), _ptr(out), n_layers, float(rule1), float(rule2), float(scale),
                  int(only_last_step))

    def step_advance(self, state):
        self.keep.append(state)
        self._add(self.lib.pv_step_advance, _ptr(state))
