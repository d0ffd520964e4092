"""SD-v1.5 ``UNet2DConditionModel`` executed by hand-written HIP kernels (gfx950).

Interface mirrored from what the reference uses of diffusers' UNet
(``/root/reference/models/infer.py:54,103-114``, ``models/unet.py:8-53``,
``models/modeling_utils.py:24,33,70``): ``unet(sample, t, encoder_hidden_states=(text, ip)).sample``,
``unet.config.{in_channels,cross_attention_dim,block_out_channels}``, ``unet.attn_processors``,
``unet.set_attn_processor``, diffusers-compatible ``state_dict`` names (checkpoint layout of
``save_progress``, ``modeling_utils.py:29-50``).

The ``nn.Module`` tree below only HOLDS parameters (same names / shapes / default init as diffusers);
none of its modules has an eager ``forward``.  Execution is a static launch plan (``UNetEngine``) over
NHWC fp16 buffers built once per input shape; see DESIGN.md.  No CPU path exists.
"""
from __future__ import annotations

import os
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .attention_processor import AttnProcessor2_0, PhotoVerseAttnProcessor2_0
from .ops import ACT_NONE, ACT_SILU, Recorder, pack_geglu, pack_geglu_rows

SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5,
)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError(f"{type(self).__name__} only holds parameters; the HIP engine executes the UNet")


class Attention(_Holder):
    """Parameter holder with the attribute surface PhotoVerse processors read from diffusers' ``Attention``
    (``attention_processor.py:275-433``)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.inner_dim, self.query_dim = heads, inner, query_dim
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.is_cross_attention = cross_attention_dim is not None
        self.spatial_norm = self.group_norm = self.norm_cross = None
        self.residual_connection, self.rescale_output_factor = False, 1.0
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(self.cross_attention_dim, inner, bias=False)
        self.to_v = nn.Linear(self.cross_attention_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.processor = AttnProcessor2_0()

    def set_processor(self, processor):
        if isinstance(getattr(self, "processor", None), nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def get_processor(self):
        return self.processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        # diffusers' Attention.forward: delegate to the installed processor (HIP-backed, standalone path)
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kw)


class GEGLU(_Holder):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(_Holder):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim)])


class BasicTransformerBlock(_Holder):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, cross_attention_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)


class Transformer2DModel(_Holder):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups=32):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Conv2d(in_channels, inner, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Conv2d(inner, in_channels, 1)


class ResnetBlock2D(_Holder):
    def __init__(self, cin, cout, temb, groups=32, eps=1e-5):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None


class Downsample2D(_Holder):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, stride=2, padding=1)


class Upsample2D(_Holder):
    def __init__(self, ch):
        super().__init__()
        self.conv = nn.Conv2d(ch, ch, 3, padding=1)


class DownBlock(_Holder):
    def __init__(self, cin, cout, temb, layers, heads, xdim, groups, has_attn, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb, groups) for i in range(layers)])
        self.has_attn = has_attn
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(heads, cout // heads, cout, xdim, groups) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None


class MidBlock(_Holder):
    def __init__(self, ch, temb, heads, xdim, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(ch, ch, temb, groups), ResnetBlock2D(ch, ch, temb, groups)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, ch // heads, ch, xdim, groups)])


class UpBlock(_Holder):
    def __init__(self, cin, cout, cprev, temb, layers, heads, xdim, groups, has_attn, add_up):
        super().__init__()
        res = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = cprev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, groups))
        self.resnets = nn.ModuleList(res)
        self.has_attn = has_attn
        if has_attn:
            self.attentions = nn.ModuleList([Transformer2DModel(heads, cout // heads, cout, xdim, groups) for _ in range(layers)])
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None


class TimestepEmbedding(_Holder):
    def __init__(self, cin, dim):
        super().__init__()
        self.linear_1 = nn.Linear(cin, dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(dim, dim)


#: A/B switch: PV_NO_XFUSED=1 runs attn2 as the four separate launches (LayerNorm, to_q GEMM, dual-branch SDPA, to_out GEMM)
USE_XFUSED = not os.environ.get("PV_NO_XFUSED")
USE_ROWGEMM = not os.environ.get("PV_NO_ROWGEMM")     # A/B switch: LayerNorm + K = 320 Linear as one row-owning launch (pv_rowgemm.hip)
GN_PROJ_IN = os.environ.get("PV_GN_PROJ_IN", "1") != "0"   # A/B switch: Transformer2DModel.norm folded into proj_in on the row-owning launch (round 5)


def _f16(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float16).contiguous()


def _f32(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _conv3_w(w: torch.Tensor) -> torch.Tensor:
    """[Cout, Cin, 3, 3] -> fp16 [Cout, 3*3*Cin] (tap-major, channel-minor: the implicit-GEMM K order)."""
    return w.detach().permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(torch.float16).contiguous()


def _conv1_w(w: torch.Tensor) -> torch.Tensor:
    return w.detach().reshape(w.shape[0], -1).to(torch.float16).contiguous()


class UNetEngine:
    """Static launch plan of one UNet forward for a fixed (batch, height, width, ip tokens, timestep rows)."""

    def __init__(self, unet: "UNet2DConditionModel", batch: int, h: int, w: int, n_ip: int, t_rows: int, device,
                 timesteps: Optional[torch.Tensor] = None, state: Optional[torch.Tensor] = None, n_text: int = 77,
                 latents_in: Optional[torch.Tensor] = None, text: Optional[torch.Tensor] = None,
                 ip: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, device_fusion: Optional[str] = None,
                 fusion_seed: int = 0, segment: Optional[str] = None, split: int = 2, mid_in=None, mid_out=None, big_min: Optional[int] = None,
                 prefix=None):
        """``device_fusion``: None - branch weights (w_text, w_ip) are launch parameters patched by the host (``_set_fusion``);
        ``"always"`` - every forward draws them on the device (grad-mode semantics of attention_processor.py:413-420, graph-safe);
        ``"last_step"`` - drawn only when the loop state says this is the last denoising step (``run_inference(training_mode=True)``,
        infer.py:99), (1, 1) otherwise.

        ``segment`` (pipeline.DenoiseLoop's low-resolution CFG merge): the forward as THREE plans.  ``"outer"`` records the first ``split``
        resolution levels of the down path into ``rec_head`` (its last launch, the level's downsampling conv, writes into ``mid_in`` = (tensor,
        column statistics): this branch's half of a buffer shared with the other CFG branch) and the matching last ``split`` up blocks + conv_out
        into ``rec_tail`` (starting from ``mid_out`` = this branch's half of the merged part's output); ``"mid"`` records everything in
        between at batch = BOTH branches (``mid_in`` / ``mid_out`` = the whole shared buffers).

        ``"prefix"`` / ``prefix=`` (optional, DenoiseLoop(share_prefix=True)): the part of a forward that does not see the conditioning - conv_in, the
        first ResnetBlock and the first transformer block up to its self-attention - is the same computation in the uncond and cond forwards of a
        CFG step (same latents, same timestep).  ``segment="prefix"`` records it once (``self.prefix_out``: conv_in's output, the ResnetBlock's
        output, the hidden states behind attn1, with the column statistics their GroupNorm consumers need); a plan given ``prefix=`` that
        dictionary starts from those tensors instead of recomputing them.  Exact: the same kernels on the same inputs."""
        self.unet, self.B, self.H, self.W, self.P, self.NT = unet, batch, h, w, n_ip, n_text
        if segment not in (None, "outer", "mid", "prefix"):
            raise ValueError("segment must be None, 'outer', 'mid' or 'prefix'")
        if segment in ("outer", "mid") and (mid_in is None or mid_out is None or device_fusion is not None):
            raise ValueError("a plan segment needs the shared mid_in / mid_out buffers (and host-side fusion weights)")
        if (segment == "prefix" or prefix is not None) and (device_fusion is not None or segment == "mid"):
            raise ValueError("the shared prefix exists for the inference plans of the two CFG branches")
        self.prefix_in, self.prefix_out = prefix, None
        self.segment, self.split, self.mid_in, self.mid_out = segment, split, mid_in, mid_out
        cfg = unet.config
        rec = self.rec = Recorder(device)
        self.rec_head, self.rec_tail = (rec, Recorder(device)) if segment == "outer" else (None, None)
        if self.rec_tail is not None:
            self.rec_tail.colstats = rec.colstats           # the skips the tail reads were written (with their statistics) by the head
        # launches that depend on the conditioning only (text / image-token K,V projections of the 16 cross-attention layers,
        # attention_processor.py:304-305,392-393): replayed when the conditioning changes, NOT every denoising step
        self.rec_cond = Recorder(device)
        if big_min is not None and "PV_CONV_BIG" not in os.environ:
            # this plan runs beside another one on a second stream: a launch of 128 one-per-CU workgroups fills ITS half of the chip
            for r in (rec, self.rec_tail):
                if r is not None:
                    r.big_min = big_min
        if segment == "mid" and "PV_CONV_BIG" not in os.environ:
            rec.big_split2 = True                           # the merged plan runs alone at batch 2B: its 16 x 16 convs take two K-slices on the one-per-CU tile
        dev = rec.device
        xdim = cfg.cross_attention_dim
        self.x_in = latents_in if latents_in is not None else rec.empty((batch, cfg.in_channels, h, w), torch.float32)
        self.text = text if text is not None else rec.empty((batch * n_text, xdim))
        self.ip = ip if ip is not None else rec.empty((batch * n_ip, xdim))
        self.timesteps = timesteps if timesteps is not None else rec.empty((t_rows,), torch.float32)
        self.state = state
        self.t_rows = t_rows
        self.out_buf = out
        self.vnorms: Dict[str, torch.Tensor] = {}
        self.xattn_params: Dict[str, object] = {}   # per-layer launch records: (w_text, w_ip) are patched per call in grad mode
        self.device_fusion = device_fusion
        self.fusion_tab = self.fusion_rng = self.fusion_forced = None
        self.fusion_names = []
        if device_fusion is not None:
            if device_fusion not in ("always", "last_step"):
                raise ValueError("device_fusion must be None, 'always' or 'last_step'")
            procs = [m.processor for _, m in unet.named_modules() if isinstance(m, Attention) and isinstance(m.processor, PhotoVerseAttnProcessor2_0)]
            nl = len(procs)
            self.fusion_tab = rec.hold(torch.ones((nl, 2), dtype=torch.float32, device=rec.device))
            import numpy as np
            key = np.array([fusion_seed & 0xFFFFFFFF, (fusion_seed >> 32) & 0xFFFFFFFF, 0, 0], dtype=np.uint32).view(np.int32)
            self.fusion_rng = rec.hold(torch.from_numpy(key.copy()).to(rec.device))      # {key lo, key hi, launch count, -}
            self.fusion_forced = rec.hold(torch.full((nl,), -1.0, dtype=torch.float32, device=rec.device))   # entries >= 0 replace the drawn u (tests)
            p0 = procs[0]
            rec.fusion_draw(self.state if device_fusion == "last_step" else None, self.fusion_rng, self.fusion_forced, self.fusion_tab, n_layers=nl,
                            rule1=p0.fusion_rule1, rule2=p0.fusion_rule2, scale=float(p0.scale[0]), only_last_step=device_fusion == "last_step")
        self._build()

    # ------------------------------------------------------------------ blocks
    def _resnet(self, m: ResnetBlock2D, x, x1, b, h, w, temb_all, toff):
        rec = self.rec
        cout = m.conv1.out_channels
        geo = dict(batch=b, hin=h, win=w, hout=h, wout=w)
        geo_t = (b, h, w, h, w, 1, 0, 1)

        def norm_conv(norm, xin, xin1, weight, **kw):
            """conv(silu(GroupNorm(xin | xin1))): where the conv runs on the LDS-resident input patch (the 64 x 64 level), the norm is folded into it -
            ONE statistics launch for the scale / shift table, no pass that writes and re-reads the normalised tensor; otherwise GroupNorm, then the conv."""
            cin0, cin1 = xin.shape[1], (xin1.shape[1] if xin1 is not None else 0)
            if Recorder.gn_conv_supported(geo_t, b * h * w, weight.shape[0], cin0, cin1, rec.big_min):
                tab = rec.groupnorm_table(xin, _f32(norm.weight), _f32(norm.bias), batch=b, hw=h * w, x1=xin1, eps=norm.eps)
                if tab is not None:
                    return rec.gemm(xin, _conv3_w(weight), a1=xin1, conv=geo, colstats=True, a_norm=tab, a_norm_act=ACT_SILU, splitk=0, **kw)
            hn = rec.groupnorm(xin, _f32(norm.weight), _f32(norm.bias), batch=b, hw=h * w, x1=xin1, eps=norm.eps, act=ACT_SILU)
            return rec.gemm(hn, _conv3_w(weight), conv=geo, colstats=True, **kw)

        h1 = norm_conv(m.norm1, x, x1, m.conv1.weight, bias=_f32(m.conv1.bias), rowadd=temb_all[:, toff:toff + cout],
                       rowadd_ld=(temb_all.stride(0) if self.t_rows > 1 else 0))
        if m.conv_shortcut is not None:
            sc = rec.gemm(x, _conv1_w(m.conv_shortcut.weight), a1=x1, bias=_f32(m.conv_shortcut.bias), rows_per_image=h * w)
        else:
            assert x1 is None
            sc = x
        # every block output feeds a GroupNorm (next norm1 / Transformer2D.norm / conv_norm_out, directly or as a skip)
        return norm_conv(m.norm2, h1, None, m.conv2.weight, bias=_f32(m.conv2.bias), residual=sc)

    def _transformer(self, name: str, m: Transformer2DModel, x, b, h, w, stop_after_attn1: bool = False, resume_hs=None):
        """``stop_after_attn1``: record norm -> proj_in -> attn1 only and return the hidden states (the conditioning-independent half of the block);
        ``resume_hs``: those hidden states, computed by another plan - record the rest (attn2, feed-forward, proj_out + residual x)."""
        rec = self.rec
        n = h * w
        blk = m.transformer_blocks[0]
        C = m.proj_in.out_channels
        heads = blk.attn1.heads
        d = C // heads
        if resume_hs is not None:
            return self._transformer_tail(name, m, x, resume_hs, b, h, w)
        tab = None
        if GN_PROJ_IN and USE_ROWGEMM and Recorder.row_gemm_supported(x.shape[1], C) and n % 128 == 0:
            # Transformer2DModel.norm folded into proj_in (64 x 64 level, K = 320): one statistics launch for the scale / shift table, then the row-owning
            # GEMM on the RAW block output - the GroupNorm-apply pass (write + re-read of the normalised tensor) disappears
            tab = rec.groupnorm_table(x, _f32(m.norm.weight), _f32(m.norm.bias), batch=b, hw=n, eps=m.norm.eps)
        if tab is not None:
            hs = rec.row_gemm(x, _conv1_w(m.proj_in.weight), bias=_f32(m.proj_in.bias), x_norm=tab, rows_per_image=n)
        else:
            g = rec.groupnorm(x, _f32(m.norm.weight), _f32(m.norm.bias), batch=b, hw=n, eps=m.norm.eps, act=ACT_NONE)
            hs = rec.gemm(g, _conv1_w(m.proj_in.weight), bias=_f32(m.proj_in.bias), rows_per_image=n)
        # --- attn1 (stock AttnProcessor2_0, models/unet.py:20-24) ---
        a1 = blk.attn1
        wqkv = torch.cat([_f16(a1.to_q.weight), _f16(a1.to_k.weight), _f16(a1.to_v.weight)], 0).contiguous()
        if USE_ROWGEMM and Recorder.row_gemm_supported(C, 3 * C):
            # norm1 + [to_q; to_k; to_v] as ONE row-owning launch (pv_rowgemm.hip): rows normalised in registers, weights streamed
            qkv = rec.row_gemm(hs, wqkv, ln_gamma=_f32(blk.norm1.weight), ln_beta=_f32(blk.norm1.bias), ln_eps=blk.norm1.eps)
        elif Recorder.gemm_ln_supported(b * n, 3 * C, C, False, rec.big_min):
            # norm1 folded into the fused qkv Linear on the 256-row tile: the GEMM reads the raw rows, the epilogue normalises
            wl, bl = Recorder.fold_layernorm(wqkv, None, _f32(blk.norm1.weight), _f32(blk.norm1.bias))
            qkv = rec.gemm(hs, wl, bias=bl, rows_per_image=n, ln_gamma=True, ln_eps=blk.norm1.eps, splitk=0)
        else:
            n1 = rec.layernorm(hs, _f32(blk.norm1.weight), _f32(blk.norm1.bias), eps=blk.norm1.eps)
            qkv = rec.gemm(n1, wqkv, rows_per_image=n)
        sa = rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=b, heads=heads, nq=n, nk=n, d=d)
        hs = rec.gemm(sa, _f16(a1.to_out[0].weight), bias=_f32(a1.to_out[0].bias), residual=hs, rows_per_image=n)
        if stop_after_attn1:
            return hs
        return self._transformer_tail(name, m, x, hs, b, h, w)

    def _transformer_tail(self, name: str, m: Transformer2DModel, x, hs, b, h, w):
        rec = self.rec
        n = h * w
        blk = m.transformer_blocks[0]
        C = m.proj_in.out_channels
        heads = blk.attn1.heads
        d = C // heads
        # --- attn2 (PhotoVerseAttnProcessor2_0, attention_processor.py:245-435) ---
        a2 = blk.attn2
        proc = a2.processor
        wkv = torch.cat([_f16(a2.to_k.weight), _f16(a2.to_v.weight)], 0).contiguous()
        kvt = self.rec_cond.gemm(self.text, wkv, rows_per_image=self.NT)
        wkvip = torch.cat([_f16(proc.to_k_ip[0].weight), _f16(proc.to_v_ip[0].weight)], 0).contiguous()
        kvip = self.rec_cond.gemm(self.ip, wkvip, rows_per_image=self.P)
        vnorm = rec.empty((b, heads, self.P), torch.float32)
        self.vnorms[name] = vnorm
        fus = None
        if self.fusion_tab is not None:                      # this layer's (w_text, w_ip) slot of the device-side draw
            fus = self.fusion_tab[len(self.fusion_names)]
            self.fusion_names.append(name)
        rec.role = f"attn2:{C}"                           # measurement label: the per-step launches of this layer's attn2 branch
        if USE_XFUSED and Recorder.xattn_fused_supported(C, heads, n, self.NT, self.P):
            # ONE launch for norm2 -> to_q -> dual-branch SDPA -> to_out + bias + residual (pv_xfused.hip); the K / V images and
            # to_v_ip_norm depend on the conditioning only (rec_cond)
            kimg, vimg = self.rec_cond.xattn_pack_kv(kvt[:, :C], kvt[:, C:], kvip[:, :C], kvip[:, C:], batch=b, heads=heads, d=d, nt=self.NT,
                                                     nip=self.P, vnorm=vnorm)
            hs, xp = rec.cross_attention_fused(hs, _f16(a2.to_q.weight), rec.pack_wo_for_fused(_f16(a2.to_out[0].weight)),
                                               _f32(a2.to_out[0].bias), kimg, vimg, batch=b, nq=n, heads=heads, d=d, nt=self.NT, nip=self.P,
                                               ln_gamma=_f32(blk.norm2.weight), ln_beta=_f32(blk.norm2.bias), ln_eps=blk.norm2.eps, fusion=fus)
            self.xattn_params[name] = xp
        elif Recorder.xattn_lnq_supported(C, heads, self.NT, self.P):
            # C = 1280 (and C = 640 when PV_XFUSED_WIDTHS leaves it out): norm2 -> to_q -> dual-branch SDPA head-parallel in one launch (pv_xq.hip), then to_out + bias + residual
            xa, xp = rec.cross_attention_lnq(hs, _f16(a2.to_q.weight), kvt[:, :C], kvt[:, C:], kvip[:, :C], kvip[:, C:], batch=b, heads=heads, nq=n,
                                             nt=self.NT, nip=self.P, d=d, ln_gamma=_f32(blk.norm2.weight), ln_beta=_f32(blk.norm2.bias),
                                             ln_eps=blk.norm2.eps, vnorm=vnorm, fusion=fus)
            self.xattn_params[name] = xp
            hs = rec.gemm(xa, _f16(a2.to_out[0].weight), bias=_f32(a2.to_out[0].bias), residual=hs, rows_per_image=n)
        else:
            n2 = rec.layernorm(hs, _f32(blk.norm2.weight), _f32(blk.norm2.bias), eps=blk.norm2.eps)
            q = rec.gemm(n2, _f16(a2.to_q.weight), rows_per_image=n)
            xa, xp = rec.cross_attention(q, kvt[:, :C], kvt[:, C:], kvip[:, :C], kvip[:, C:], batch=b, heads=heads, nq=n, nt=self.NT,
                                         nip=self.P, d=d, vnorm=vnorm, fusion=fus)
            self.xattn_params[name] = xp
            hs = rec.gemm(xa, _f16(a2.to_out[0].weight), bias=_f32(a2.to_out[0].bias), residual=hs, rows_per_image=n)
        rec.role = None
        # --- GEGLU feed-forward ---
        if USE_ROWGEMM and Recorder.row_gemm_supported(C, blk.ff.net[0].proj.weight.shape[0]):
            # norm3 + GEGLU projection + gate as ONE row-owning launch
            wg, bg = pack_geglu_rows(_f16(blk.ff.net[0].proj.weight), _f32(blk.ff.net[0].proj.bias))
            gg = rec.row_gemm(hs, wg, bias=bg, ln_gamma=_f32(blk.norm3.weight), ln_beta=_f32(blk.norm3.bias), ln_eps=blk.norm3.eps, geglu=True)
        elif Recorder.gemm_ln_supported(b * n, blk.ff.net[0].proj.weight.shape[0], C, True, rec.big_min):
            # norm3 folded into the GEGLU projection (256-row tile, 256-column tiles)
            wl, bl = Recorder.fold_layernorm(_f16(blk.ff.net[0].proj.weight), _f32(blk.ff.net[0].proj.bias), _f32(blk.norm3.weight), _f32(blk.norm3.bias))
            wg, bg = pack_geglu(wl, bl)
            gg = rec.gemm(hs, wg, bias=bg, geglu=True, rows_per_image=n, ln_gamma=True, ln_eps=blk.norm3.eps)
        else:
            n3 = rec.layernorm(hs, _f32(blk.norm3.weight), _f32(blk.norm3.bias), eps=blk.norm3.eps)
            wg, bg = pack_geglu(_f16(blk.ff.net[0].proj.weight), _f32(blk.ff.net[0].proj.bias))
            gg = rec.gemm(n3, wg, bias=bg, geglu=True, rows_per_image=n)
        hs = rec.gemm(gg, _f16(blk.ff.net[2].weight), bias=_f32(blk.ff.net[2].bias), residual=hs, rows_per_image=n)
        return rec.gemm(hs, _conv1_w(m.proj_out.weight), bias=_f32(m.proj_out.bias), residual=x, rows_per_image=n, colstats=True)

    # ------------------------------------------------------------------ plan
    def _build(self):
        u, rec, B = self.unet, self.rec, self.B
        cfg = u.config
        h, w = self.H, self.W
        c0 = cfg.block_out_channels[0]
        # time embedding: sinusoid -> linear_1+SiLU -> linear_2 (+SiLU, the only consumer is time_emb_proj(silu(emb)))
        te = rec.timestep_embedding(self.timesteps, self.state, self.t_rows, c0)
        e1 = rec.gemm(te, _f16(u.time_embedding.linear_1.weight), bias=_f32(u.time_embedding.linear_1.bias), act=ACT_SILU)
        e2 = rec.gemm(e1, _f16(u.time_embedding.linear_2.weight), bias=_f32(u.time_embedding.linear_2.bias), act=ACT_SILU)
        resnets = [m for m in u.modules() if isinstance(m, ResnetBlock2D)]
        toffs, off = {}, 0
        for m in resnets:
            toffs[id(m)] = off
            off += m.conv1.out_channels
        pad = (-off) % 160
        wt = torch.cat([_f16(m.time_emb_proj.weight) for m in resnets] +
                       ([torch.zeros(pad, resnets[0].time_emb_proj.in_features, dtype=torch.float16, device=rec.device)] if pad else []), 0)
        bt = torch.cat([_f32(m.time_emb_proj.bias) for m in resnets] + ([torch.zeros(pad, device=rec.device)] if pad else []), 0)
        temb_all = rec.gemm(e2, wt.contiguous(), bias=bt.contiguous(), out_f32=True)

        seg, split = self.segment, self.split
        n_lv = len(u.down_blocks)
        if seg in ("outer", "mid") and not (0 < split < n_lv):
            raise ValueError("split must leave at least one resolution level on either side")

        def adopt(t_cs, rows):
            """a tensor written by ANOTHER plan (with its column statistics) becomes this plan's current activation"""
            t, cs = t_cs
            assert t.shape[0] == rows
            key = (t.data_ptr(), rows, t.shape[1])
            # statistics are adopted only when the producing plans really write them: with the PV_NO_COLSTATS A/B switch (ops._NO_COLSTATS)
            # Recorder.gemm leaves a caller-owned statistics buffer untouched, and GroupNorm must fall back to its own statistics pass
            # instead of normalising with the zero-initialised buffer
            if cs is not None and not ops._NO_COLSTATS:
                self.rec.colstats[key] = cs
            else:
                self.rec.colstats.pop(key, None)
            return t

        pre = self.prefix_in
        first = u.down_blocks[0]
        if (seg == "prefix" or pre is not None) and not (first.has_attn and len(first.resnets) >= 1):
            raise ValueError("the shared prefix needs an attention down block first (conv_in -> ResnetBlock -> transformer block)")
        if pre is not None:
            # conv_in, the first ResnetBlock and the first transformer block up to attn1 were recorded by the prefix plan
            x = adopt(pre["conv_in"], B * h * w)
            skips = [(x, h, w)]
        elif seg != "mid":
            # conv_in (cin = 4): im2col to K = 36 -> 64 (zero padded), then the MFMA GEMM
            kin = cfg.in_channels * 9
            kpad = (kin + 63) // 64 * 64
            cols = rec.im2col3x3(self.x_in, batch=B, cin=cfg.in_channels, h=h, wd=w, kpad=kpad)
            w_in = torch.zeros(c0, kpad, dtype=torch.float16, device=rec.device)
            w_in[:, :kin] = u.conv_in.weight.detach().reshape(c0, kin).to(torch.float16)
            x = rec.gemm(cols, w_in, bias=_f32(u.conv_in.bias), rows_per_image=h * w, colstats=True)
            skips = [(x, h, w)]
            if seg == "prefix":
                key = lambda t: (t.data_ptr(), t.shape[0], t.shape[1])
                res0 = first.resnets[0]
                xr = self._resnet(res0, x, None, B, h, w, temb_all, toffs[id(res0)])
                hs = self._transformer("down_blocks.0.attentions.0", first.attentions[0], xr, B, h, w, stop_after_attn1=True)
                self.prefix_out = {"conv_in": (x, rec.colstats.get(key(x))), "res": (xr, rec.colstats.get(key(xr))), "hs": hs}
                self.out = hs
                return
        else:
            h, w = h >> split, w >> split
            x = adopt(self.mid_in, B * h * w)
            skips = [(x, h, w)]
        for bi, blk in enumerate(u.down_blocks):
            if (seg == "outer" and bi >= split) or (seg == "mid" and bi < split):
                continue
            for i, res in enumerate(blk.resnets):
                if pre is not None and bi == 0 and i == 0:
                    x = adopt(pre["res"], B * h * w)
                    x = self._transformer("down_blocks.0.attentions.0", blk.attentions[0], x, B, h, w, resume_hs=pre["hs"])
                    skips.append((x, h, w))
                    continue
                x = self._resnet(res, x, None, B, h, w, temb_all, toffs[id(res)])
                if blk.has_attn:
                    x = self._transformer(f"down_blocks.{bi}.attentions.{i}", blk.attentions[i], x, B, h, w)
                skips.append((x, h, w))
            if blk.downsamplers is not None:
                conv = blk.downsamplers[0].conv
                boundary = seg == "outer" and bi == split - 1         # this branch's half of the merged part's input
                x = rec.gemm(x, _conv3_w(conv.weight), bias=_f32(conv.bias),
                             conv=dict(batch=B, hin=h, win=w, hout=h // 2, wout=w // 2, stride=2), colstats=True,
                             out=self.mid_in[0] if boundary else None, colstats_out=self.mid_in[1] if boundary else None)
                h, w = h // 2, w // 2
                if not boundary:
                    skips.append((x, h, w))                           # (at the boundary the skip belongs to the merged part)
        if seg != "outer":
            mb = u.mid_block
            x = self._resnet(mb.resnets[0], x, None, B, h, w, temb_all, toffs[id(mb.resnets[0])])
            x = self._transformer("mid_block.attentions.0", mb.attentions[0], x, B, h, w)
            x = self._resnet(mb.resnets[1], x, None, B, h, w, temb_all, toffs[id(mb.resnets[1])])
        n_up = len(u.up_blocks)
        for bi, blk in enumerate(u.up_blocks):
            outer_up = bi >= n_up - split
            if (seg == "outer" and not outer_up) or (seg == "mid" and outer_up):
                continue
            if seg == "outer" and bi == n_up - split:
                rec = self.rec = self.rec_tail                        # the tail plan starts from this branch's half of the merged output
                h, w = self.H >> (split - 1), self.W >> (split - 1)
                x = adopt(self.mid_out, B * h * w)
            for i, res in enumerate(blk.resnets):
                sk, sh, sw = skips.pop()
                assert (sh, sw) == (h, w)
                x = self._resnet(res, x, sk, B, h, w, temb_all, toffs[id(res)])   # channel concat [x | skip] is never materialised
                if blk.has_attn:
                    x = self._transformer(f"up_blocks.{bi}.attentions.{i}", blk.attentions[i], x, B, h, w)
            if blk.upsamplers is not None:
                conv = blk.upsamplers[0].conv
                boundary = seg == "mid" and bi == n_up - split - 1
                x = rec.gemm(x, _conv3_w(conv.weight), bias=_f32(conv.bias),
                             conv=dict(batch=B, hin=h, win=w, hout=h * 2, wout=w * 2, upsample=1), colstats=True,
                             out=self.mid_out[0] if boundary else None, colstats_out=self.mid_out[1] if boundary else None)
                h, w = h * 2, w * 2
        assert not skips, "every skip connection is consumed inside the plan segment that produced it"
        if seg == "mid":
            self.out = x
            return
        xn = rec.groupnorm(x, _f32(u.conv_norm_out.weight), _f32(u.conv_norm_out.bias), batch=B, hw=h * w,
                           eps=u.conv_norm_out.eps, act=ACT_SILU)
        wo = u.conv_out.weight.detach().permute(0, 2, 3, 1).reshape(cfg.out_channels, -1).to(torch.float16).contiguous()
        self.out = rec.conv_out(xn, wo, _f32(u.conv_out.bias), batch=B, cin=c0, h=h, wd=w, cout=cfg.out_channels, out=self.out_buf)
        if seg == "outer":
            self.rec = Recorder.concat(self.rec_head, self.rec_tail)      # introspection / timing view of the branch's own launches

    def run_conditioning(self):
        self.rec_cond.run()

    def run(self, conditioning=True):
        if conditioning:
            self.rec_cond.run()
        self.rec.run()
        return self.out


class UNet2DConditionModel(nn.Module):
    def __init__(self, **overrides):
        super().__init__()
        cfg = dict(SD15_CONFIG)
        cfg.update(overrides)
        self.config = SimpleNamespace(**cfg)
        boc = tuple(cfg["block_out_channels"])
        heads, xdim, groups, layers = cfg["attention_head_dim"], cfg["cross_attention_dim"], cfg["norm_num_groups"], cfg["layers_per_block"]
        temb = boc[0] * 4
        self.conv_in = nn.Conv2d(cfg["in_channels"], boc[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(boc[0], temb)
        self.down_blocks = nn.ModuleList()
        cout = boc[0]
        for i, kind in enumerate(cfg["down_block_types"]):
            cin, cout = cout, boc[i]
            self.down_blocks.append(DownBlock(cin, cout, temb, layers, heads, xdim, groups, kind.startswith("CrossAttn"), i != len(boc) - 1))
        self.mid_block = MidBlock(boc[-1], temb, heads, xdim, groups)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        cout = rev[0]
        for i, kind in enumerate(cfg["up_block_types"]):
            cprev, cout = cout, rev[i]
            cin = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(UpBlock(cin, cout, cprev, temb, layers + 1, heads, xdim, groups, kind.startswith("CrossAttn"), i != len(boc) - 1))
        self.conv_norm_out = nn.GroupNorm(groups, boc[0], eps=cfg["norm_eps"])
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)
        self._engines: Dict[tuple, UNetEngine] = {}

    # ---- diffusers processor plumbing (models/unet.py:8-47) ----
    @property
    def attn_processors(self) -> Dict[str, object]:
        return {f"{n}.processor": m.get_processor() for n, m in self.named_modules() if isinstance(m, Attention)}

    def set_attn_processor(self, processor):
        for n, m in self.named_modules():
            if isinstance(m, Attention):
                m.set_processor(processor[f"{n}.processor"] if isinstance(processor, dict) else processor)
        self.repack()

    def repack(self):
        """Drop cached launch plans (call after changing weights / processors)."""
        self._engines.clear()
        self.__dict__["_pack_version"] = self.__dict__.get("_pack_version", 0) + 1
        self.__dict__.pop("_denoise_loops", None)

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.repack()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._engines = {}
        self.__dict__["_pack_version"] = self.__dict__.get("_pack_version", 0) + 1
        self.__dict__.pop("_denoise_loops", None)
        return r

    @property
    def device(self):
        return self.conv_in.weight.device

    def engine(self, batch, h, w, n_ip, t_rows, **kw) -> UNetEngine:
        """Launch plan for one input shape.  With ``kw`` (caller-owned static buffers: ``latents_in``, ``text``, ``ip``,
        ``timesteps``, ``state``) a private, uncached engine is built - used by the graph-captured denoise loop."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("photoverse_amd.UNet2DConditionModel runs on a HIP device only (no CPU path): call .to('cuda')")
        for proc in self.attn_processors.values():
            if not isinstance(proc, (AttnProcessor2_0, PhotoVerseAttnProcessor2_0)):
                raise TypeError(f"unsupported attention processor {type(proc).__name__}")
        if kw:
            return UNetEngine(self, batch, h, w, n_ip, t_rows, dev, **kw)
        key = (batch, h, w, n_ip, t_rows)
        eng = self._engines.get(key)
        if eng is None:
            eng = self._engines[key] = UNetEngine(self, batch, h, w, n_ip, t_rows, dev)
        return eng

    def _set_fusion(self, eng: UNetEngine):
        """Per-layer branch weights: (1,1) under no_grad (attention_processor.py:411-412); in grad mode every cross-attn
        layer draws its own u~U(0,1) per call (:413-420) - host-side, like the reference's ``torch.rand(1).item()``."""
        if eng.device_fusion is not None:
            return
        for name, m in self.named_modules():
            if isinstance(m, Attention) and isinstance(m.processor, PhotoVerseAttnProcessor2_0):
                xp = eng.xattn_params[name.rsplit(".transformer_blocks", 1)[0]]
                xp.w_text, xp.w_ip = m.processor.branch_weights()

    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states=None):
        ops.require_cuda(sample, "sample")
        if not isinstance(encoder_hidden_states, tuple):
            raise TypeError("encoder_hidden_states must be the (text, ip) tuple of attention_processor.py:258-262")
        text, ip = encoder_hidden_states
        if isinstance(ip, list):
            ip = ip[0]
        B, _, h, w = sample.shape
        t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep])
        t = t.reshape(-1).to(device=sample.device, dtype=torch.float32)
        eng = self.engine(B, h, w, ip.shape[1], t.numel())
        if eng.NT != text.shape[1]:
            raise ValueError(f"text sequence length {text.shape[1]} != {eng.NT}")
        eng.x_in.copy_(sample)
        eng.text.copy_(text.reshape(-1, text.shape[-1]))
        eng.ip.copy_(ip.reshape(-1, ip.shape[-1]))
        eng.timesteps.copy_(t)
        self._set_fusion(eng)
        out = eng.run()
        # side output of the PhotoVerse processors (attention_processor.py:397)
        for name, m in self.named_modules():
            if isinstance(m, Attention) and isinstance(m.processor, PhotoVerseAttnProcessor2_0):
                key = name.rsplit(".transformer_blocks", 1)[0]
                m.processor.to_v_ip_norm = eng.vnorms[key].view(B, m.heads, -1, 1)
        return SimpleNamespace(sample=out.clone().to(sample.dtype))


def set_visual_cross_attention_adapter(unet, num_tokens=(5,)):
    """Same contract as ``/root/reference/models/unet.py:8-35``: attn1 -> stock processor, attn2 ->
    PhotoVerse processor sized from the block name."""
    procs = {}
    boc = unet.config.block_out_channels
    for name in unet.attn_processors.keys():
        cross_attention_dim = None if name.endswith("attn1.processor") else unet.config.cross_attention_dim
        if name.startswith("mid_block"):
            hidden_size = boc[-1]
        elif name.startswith("up_blocks"):
            hidden_size = list(reversed(boc))[int(name[len("up_blocks.")])]
        elif name.startswith("down_blocks"):
            hidden_size = boc[int(name[len("down_blocks.")])]
        if cross_attention_dim is None:
            procs[name] = AttnProcessor2_0()
        else:
            procs[name] = PhotoVerseAttnProcessor2_0(cross_attention_dim=cross_attention_dim, hidden_size=hidden_size, num_tokens=num_tokens)
    unet.set_attn_processor(procs)
    dev = unet.device
    for p in procs.values():
        if isinstance(p, nn.Module):
            p.to(dev)
    return unet


def get_visual_cross_attention_values_norm(unet):
    """``/root/reference/models/unet.py:38-47``."""
    vals = [p.to_v_ip_norm for n, p in unet.attn_processors.items() if not n.endswith("attn1.processor")]
    out = torch.stack(vals, dim=1)
    return out.view(out.shape[0], -1)


def set_cross_attention_layers_to_train(unet):
    """``/root/reference/models/unet.py:50-53``."""
    for name, module in unet.named_modules():
        if "attn2" in name:
            module.train()
