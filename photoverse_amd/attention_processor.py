"""HIP-backed attention processors with the reference's plug-in protocol.

* ``AttnProcessor2_0``            - stock self-attention processor the reference installs on ``attn1``
  (``/root/reference/models/unet.py:20-24``; [EXT] diffusers).
* ``PhotoVerseAttnProcessor2_0``  - ``/root/reference/models/attention_processor.py:221-435``: text cross-attention
  plus an image-token cross-attention with its own ``to_k_ip`` / ``to_v_ip``, two independent softmaxes,
  summed (no_grad) or randomly fused (grad mode), shared ``to_out``; side output ``to_v_ip_norm``.

Protocol: ``processor(attn, hidden_states, encoder_hidden_states=(text, ip) , ...) -> Tensor`` of the shape of
``hidden_states``; ``attn`` provides ``to_q/to_k/to_v/to_out/heads``.  Calling a processor directly runs its own small
launch list (q/kv GEMMs -> fused dual-branch attention kernel -> out GEMM); inside ``UNet2DConditionModel`` the same
launches are part of the UNet's static plan.  The legacy bmm processor (``:12-218``) is only selected when torch lacks
SDPA (``unet.py:26-28``) and has different semantics; it is documented, not built.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn

from .ops import Recorder, require_cuda


def _as_f16_rows(rec: Recorder, t: torch.Tensor) -> torch.Tensor:
    t2 = t.reshape(-1, t.shape[-1])
    if t2.dtype == torch.float16:
        return rec.hold(t2.contiguous())
    return rec.cast_to_f16(rec.hold(t2.contiguous().float()))


def _w16(lin: nn.Linear) -> torch.Tensor:
    return lin.weight.detach().to(torch.float16).contiguous()


def _finish(rec: Recorder, out16: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    res = out16 if like.dtype == torch.float16 else rec.cast_to_f32(out16)
    rec.run()
    return res.view(like.shape).to(like.dtype)


class _PhotoVerseAttn2Fn(torch.autograd.Function):
    """The attn2 branch of ``PhotoVerseAttnProcessor2_0.__call__`` (attention_processor.py:297-423) as an autograd node whose forward
    AND backward run on the HIP kernels: the reference trains exactly these pieces of its own code - ``to_k_ip`` / ``to_v_ip``, the
    LoRA factors behind ``to_q`` / ``to_k`` / ``to_v`` (via the merged weight) - and needs d/d(hidden), d/d(text), d/d(ip) to keep
    back-propagating into the UNet and the adapters.  Outputs: (hidden_out, to_v_ip_norm) - the norm is differentiable too
    (train.py:512-513 regularises its mean).

    Backward (pv_backward.hip + pv_gemm_conv): dctx = dout Wo; (dq, dK_t, dV_t, dK_ip, dV_ip) = SDPA backward; dX = dY W through
    the MFMA GEMM with transposed weights; dW = dY^T X through the same GEMM on transposed operands.  ``grad_scale`` multiplies the
    incoming gradient before it is rounded to fp16 and is divided out of every result (loss-scaling, fp16 gradient storage)."""

    @staticmethod
    def forward(ctx, hidden, text, ip, wq, wk, wv, wo, bo, wkip, wvip, heads, w_text, w_ip, grad_scale):
        B, N, C = hidden.shape
        d = C // heads
        rec = Recorder(hidden.device)
        x16, t16, i16 = _as_f16_rows(rec, hidden.detach()), _as_f16_rows(rec, text.detach()), _as_f16_rows(rec, ip.detach())
        w16 = [w.detach().to(torch.float16).contiguous() for w in (wq, wk, wv, wo, wkip, wvip)]
        q = rec.gemm(x16, w16[0], rows_per_image=N)
        kvt = rec.gemm(t16, torch.cat([w16[1], w16[2]], 0).contiguous(), rows_per_image=text.shape[1])
        kvip = rec.gemm(i16, torch.cat([w16[4], w16[5]], 0).contiguous(), rows_per_image=ip.shape[1])
        vnorm = rec.empty((B, heads, ip.shape[1]), torch.float32)
        o, _ = rec.cross_attention(q, kvt[:, :C], kvt[:, C:], kvip[:, :C], kvip[:, C:], batch=B, heads=heads, nq=N, nt=text.shape[1],
                                   nip=ip.shape[1], d=d, w_text=w_text, w_ip=w_ip, vnorm=vnorm)
        out = rec.gemm(o, w16[3], bias=None if bo is None else bo.detach().float().contiguous(), rows_per_image=N)
        res = out if hidden.dtype == torch.float16 else rec.cast_to_f32(out)
        rec.run()
        ctx.save_for_backward(x16, t16, i16, q, kvt, kvip, o, *w16)
        ctx.meta = (B, N, C, heads, d, text.shape[1], ip.shape[1], float(w_text), float(w_ip), float(grad_scale), hidden.dtype, text.dtype, ip.dtype,
                    bo is not None)
        return res.view(B, N, C).to(hidden.dtype), vnorm.view(B, heads, -1, 1)

    @staticmethod
    def backward(ctx, dout, dvnorm):
        x16, t16, i16, q, kvt, kvip, o, wq, wk, wv, wo, wkip, wvip = ctx.saved_tensors
        B, N, C, heads, d, nt, nip, w_text, w_ip, S, hdt, tdt, idt, has_bias = ctx.meta
        need = ctx.needs_input_grad
        rec = Recorder(x16.device)
        inv = 1.0 / S
        # incoming gradient, loss-scaled and rounded to fp16 (zero when only the norm output is used)
        if dout is None:
            dout = torch.zeros((B, N, C), dtype=torch.float32, device=x16.device)
        scale_vec = torch.full((B,), S, dtype=torch.float32, device=x16.device)
        d32 = rec.affine_rows(dout.detach().reshape(B, -1).float().contiguous(), scale_vec)
        d16 = rec.cast_to_f16(d32.view(B * N, C))
        dctx = rec.gemm(d16, wo.t().contiguous(), rows_per_image=N)                                    # dL/d(SDPA output) = dout . Wo
        vg = None
        if dvnorm is not None:
            vg = rec.affine_rows(dvnorm.detach().reshape(B, -1).float().contiguous(), scale_vec)      # same loss scale as the main path
        dq, dkv_t32, dkv_i32 = rec.cross_attention_backward(q, kvt[:, :C], kvt[:, C:], kvip[:, :C], kvip[:, C:], dctx, batch=B, heads=heads,
                                                                 nq=N, nt=nt, nip=nip, d=d, w_text=w_text, w_ip=w_ip, vnorm_grad=vg)
        grads = [None] * 14
        pend = {}
        if need[0]:
            pend["hidden"] = rec.gemm(dq, wq.t().contiguous(), out_f32=True, rows_per_image=N)            # [B*N, C_in]
        dkv_t = rec.cast_to_f16(dkv_t32) if (need[1] or need[4] or need[5]) else None     # [dK_t | dV_t] as the next GEMMs' fp16 operand
        dkv_i = rec.cast_to_f16(dkv_i32) if (need[2] or need[8] or need[9]) else None
        if need[1]:
            pend["text"] = rec.gemm(dkv_t, torch.cat([wk, wv], 0).t().contiguous(), out_f32=True, rows_per_image=nt)
        if need[2]:
            pend["ip"] = rec.gemm(dkv_i, torch.cat([wkip, wvip], 0).t().contiguous(), out_f32=True, rows_per_image=nip)
        if need[3]:
            pend["wq"] = rec.wgrad(dq, x16)
        if need[4]:
            pend["wk"] = rec.wgrad(dkv_t[:, :C], t16)
        if need[5]:
            pend["wv"] = rec.wgrad(dkv_t[:, C:], t16)
        if need[6]:
            pend["wo"] = rec.wgrad(d16, o)
        if need[7] and has_bias:
            pend["bo"] = rec.colsum(d16)
        if need[8]:
            pend["wkip"] = rec.wgrad(dkv_i[:, :C], i16)
        if need[9]:
            pend["wvip"] = rec.wgrad(dkv_i[:, C:], i16)
        # un-scale every result with the row-affine kernel (one "row" per tensor)
        outs = {}
        one = torch.full((1,), inv, dtype=torch.float32, device=x16.device)
        for k, t in pend.items():
            outs[k] = rec.affine_rows(t.reshape(1, -1), one).view(t.shape)
        rec.run()
        if "hidden" in outs: grads[0] = outs["hidden"].view(B, N, -1).to(hdt)
        if "text" in outs: grads[1] = outs["text"].view(B, nt, -1).to(tdt)
        if "ip" in outs: grads[2] = outs["ip"].view(B, nip, -1).to(idt)
        for idx, k in ((3, "wq"), (4, "wk"), (5, "wv"), (6, "wo"), (7, "bo"), (8, "wkip"), (9, "wvip")):
            if k in outs:
                grads[idx] = outs[k]
        return tuple(grads)


class AttnProcessor2_0:
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, **kw):
        require_cuda(hidden_states, "hidden_states")
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is never passed on the PhotoVerse path")
        B, N, C = hidden_states.shape
        rec = Recorder(hidden_states.device)
        x = _as_f16_rows(rec, hidden_states)
        ctx = x if encoder_hidden_states is None else _as_f16_rows(rec, encoder_hidden_states)
        nk = N if encoder_hidden_states is None else encoder_hidden_states.shape[1]
        heads = attn.heads
        d = attn.to_q.out_features // heads
        q = rec.gemm(x, _w16(attn.to_q), rows_per_image=N)
        kv = rec.gemm(ctx, torch.cat([_w16(attn.to_k), _w16(attn.to_v)], 0).contiguous(), rows_per_image=nk)
        inner = heads * d
        o = rec.attention(q, kv[:, :inner], kv[:, inner:], batch=B, heads=heads, nq=N, nk=nk, d=d)
        out = rec.gemm(o, _w16(attn.to_out[0]), bias=attn.to_out[0].bias.detach().float().contiguous(), rows_per_image=N)
        return _finish(rec, out, hidden_states)


class PhotoVerseAttnProcessor2_0(nn.Module):
    def __init__(self, hidden_size, cross_attention_dim=None, num_tokens=(5,), scale=2.0, fusion_rules=(1 / 3, 2 / 3)):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        if not isinstance(num_tokens, (tuple, list)):
            num_tokens = [num_tokens]
        self.num_tokens = num_tokens
        # same validation and messages as attention_processor.py:37-48
        if not isinstance(fusion_rules, tuple) or len(fusion_rules) != 2 or not all(isinstance(i, float) for i in fusion_rules):
            raise ValueError("`fusion_rules` should be a tuple of two floats.")
        self.fusion_rule1, self.fusion_rule2 = fusion_rules
        if self.fusion_rule1 + self.fusion_rule2 != 1:
            raise ValueError("Sum of the fusion rules should be equal to 1.")
        if not isinstance(scale, list):
            scale = [scale] * len(num_tokens)
        if len(scale) != len(num_tokens):
            raise ValueError("`scale` should be a list of integers with the same length as `num_tokens`.")
        self.scale = scale
        self.to_k_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in range(len(num_tokens))])
        self.to_v_ip = nn.ModuleList([nn.Linear(cross_attention_dim, hidden_size, bias=False) for _ in range(len(num_tokens))])
        self.to_v_ip_norm = None
        #: test hook replacing ``torch.rand(1).item()`` of attention_processor.py:414
        self.forced_fusion_seed: Optional[float] = None

    def branch_weights(self) -> Tuple[float, float]:
        """(w_text, w_ip): (1,1) under no_grad (:411-412); in grad mode u~U(0,1): u<r1 -> (scale,0),
        u>r2 -> (0,scale), else (1,1) (:413-420).  The draw uses the CPU global generator like the reference."""
        if not torch.is_grad_enabled():
            return (1.0, 1.0)
        seed = torch.rand(1).item() if self.forced_fusion_seed is None else self.forced_fusion_seed
        s = float(self.scale[0])
        if seed < self.fusion_rule1:
            return (s, 0.0)
        if seed > self.fusion_rule2:
            return (0.0, s)
        return (1.0, 1.0)

    def split_encoder_hidden_states(self, encoder_hidden_states):
        """attention_processor.py:258-273: tuple convention, list form, deprecated bare tensor."""
        if isinstance(encoder_hidden_states, tuple):
            text, ip = encoder_hidden_states
            if isinstance(ip, list):
                if len(ip) != 1:
                    raise ValueError("PhotoVerse installs exactly one image-token branch (len(num_tokens) == 1)")
                ip = ip[0]
            return text, ip
        end_pos = encoder_hidden_states.shape[1] - self.num_tokens[0]
        return encoder_hidden_states[:, :end_pos, :], encoder_hidden_states[:, end_pos:, :]

    def forward(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, scale=2.0,
                ip_adapter_masks=None):
        require_cuda(hidden_states, "hidden_states")
        if ip_adapter_masks is not None or attention_mask is not None:
            raise NotImplementedError("mask branches (attention_processor.py:324-390) are dead for PhotoVerse callers")
        if encoder_hidden_states is None:
            raise ValueError("PhotoVerseAttnProcessor2_0 needs encoder_hidden_states=(text, ip)")
        text, ip = self.split_encoder_hidden_states(encoder_hidden_states)
        B, N, C = hidden_states.shape
        heads = attn.heads
        if torch.is_grad_enabled():
            # grad mode: the same launches as an autograd node with a HIP backward (the fusion draw happens here, like :413-420)
            wt, wi = self.branch_weights()
            bo = attn.to_out[0].bias
            out, vnorm = _PhotoVerseAttn2Fn.apply(hidden_states, text, ip, attn.to_q.weight, attn.to_k.weight, attn.to_v.weight,
                                                  attn.to_out[0].weight, bo, self.to_k_ip[0].weight, self.to_v_ip[0].weight, heads, wt, wi,
                                                  float(getattr(self, "grad_scale", 1.0)))
            self.to_v_ip_norm = vnorm                                                           # :397 (differentiable)
            return out
        d = attn.to_q.out_features // heads
        inner = heads * d
        rec = Recorder(hidden_states.device)
        x = _as_f16_rows(rec, hidden_states)
        q = rec.gemm(x, _w16(attn.to_q), rows_per_image=N)
        kvt = rec.gemm(_as_f16_rows(rec, text), torch.cat([_w16(attn.to_k), _w16(attn.to_v)], 0).contiguous(), rows_per_image=text.shape[1])
        kvip = rec.gemm(_as_f16_rows(rec, ip), torch.cat([_w16(self.to_k_ip[0]), _w16(self.to_v_ip[0])], 0).contiguous(),
                        rows_per_image=ip.shape[1])
        vnorm = rec.empty((B, heads, ip.shape[1]), torch.float32)
        wt, wi = self.branch_weights()
        o, _ = rec.cross_attention(q, kvt[:, :inner], kvt[:, inner:], kvip[:, :inner], kvip[:, inner:], batch=B, heads=heads, nq=N,
                                   nt=text.shape[1], nip=ip.shape[1], d=d, w_text=wt, w_ip=wi, vnorm=vnorm)
        out = rec.gemm(o, _w16(attn.to_out[0]), bias=attn.to_out[0].bias.detach().float().contiguous(), rows_per_image=N)
        res = _finish(rec, out, hidden_states)
        self.to_v_ip_norm = vnorm.view(B, heads, -1, 1)   # :397
        return res
