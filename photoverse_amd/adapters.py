"""``PhotoVerseAdapter`` (``/root/reference/models/adapters.py:5-44``) on HIP kernels.

Per token i: ``mapping_i(emb_i[:, :1]) + mean_over_patches(mapping_patch_i(emb_i[:, 1:]))`` where each mapping is
Linear(1024)-LN-LeakyReLU-Linear(1024)-LN-LeakyReLU-Linear(768).  ``token_index=int`` selects one embedding / mapping and
returns (B,1,768); ``None`` / ``'full'`` concatenates all tokens.  Same parameter names as the reference (checkpoint keys
``image_adapter`` / ``text_adapter``, ``modeling_utils.py:29-50``).  LayerNorm+LeakyReLU is one kernel; the patch mean
is accumulated onto the CLS mapping by ``pv_rows_mean``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .ops import ACT_LEAKY_RELU, Recorder, require_cuda


class _AddFn(torch.autograd.Function):
    """a + b on the HIP row-affine kernel (gradient: pass-through to both)."""

    @staticmethod
    def forward(ctx, a, b):
        rec = Recorder(a.device)
        one = torch.ones((1,), dtype=torch.float32, device=a.device)
        out = rec.affine_rows(a.detach().contiguous().view(1, -1), one, b.detach().contiguous().view(1, -1), one).view(a.shape)
        rec.run()
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class _CatFn(torch.autograd.Function):
    """torch.cat(tokens, dim=1) (adapters.py:43): a copy, no arithmetic; gradient: the matching slices."""

    @staticmethod
    def forward(ctx, *tokens):
        ctx.n = len(tokens)
        return torch.cat([t.detach() for t in tokens], dim=1)

    @staticmethod
    def backward(ctx, g):
        return tuple(g[:, i:i + 1].contiguous() for i in range(ctx.n))


def _mapping(cin, cout):
    return nn.Sequential(nn.Linear(cin, 1024), nn.LayerNorm(1024), nn.LeakyReLU(),
                         nn.Linear(1024, 1024), nn.LayerNorm(1024), nn.LeakyReLU(),
                         nn.Linear(1024, cout))


def _w(lin):
    return lin.weight.detach().to(torch.float16).contiguous(), lin.bias.detach().float().contiguous()


class _AdapterMLPFn(torch.autograd.Function):
    """One mapping MLP of the adapter (adapters.py:14-28: Linear-LN-LeakyReLU-Linear-LN-LeakyReLU-Linear), optionally followed by
    the patch-token mean of ``adapters.py:36`` (``group`` = tokens per sample, the first one - the CLS row - left out), as an autograd
    node whose forward and backward run on the HIP kernels.  The adapters are what PhotoVerse trains (train.py:372-377); this is
    their backward: dW = dY^T X and dX = dY W on the MFMA GEMM, LayerNorm + LeakyReLU backward in ``pv_layernorm_backward``, bias
    gradients as deterministic column sums.  ``grad_scale``: loss scaling for the fp16 gradient operands."""

    @staticmethod
    def forward(ctx, x, w0, b0, g1, e1, w3, b3, g4, e4, w6, b6, eps1, eps4, group, grad_scale):
        rec = Recorder(x.device)
        M = x.shape[0]
        x16 = x.detach() if (x.dtype == torch.float16 and x.stride(-1) == 1) else x.detach().to(torch.float16).contiguous()
        w0h, w3h, w6h = (w.detach().to(torch.float16).contiguous() for w in (w0, w3, w6))
        f = lambda t: t.detach().float().contiguous()
        h0 = rec.gemm(x16, w0h, bias=f(b0))
        a1 = rec.layernorm(h0, f(g1), f(e1), eps=eps1, act=ACT_LEAKY_RELU)
        h3 = rec.gemm(a1, w3h, bias=f(b3))
        a4 = rec.layernorm(h3, f(g4), f(e4), eps=eps4, act=ACT_LEAKY_RELU)
        y = rec.gemm(a4, w6h, bias=f(b6))
        out = y
        if group > 1:
            out = rec.rows_mean(y[1:], groups=M // group, count=group - 1, group_rows=group)
        res = rec.cast_to_f32(out)
        rec.run()
        ctx.save_for_backward(x16, h0, a1, h3, a4, w0h, w3h, w6h, f(g1), f(e1), f(g4), f(e4))
        ctx.meta = (M, int(group), float(eps1), float(eps4), float(grad_scale))
        return res

    @staticmethod
    def backward(ctx, dy):
        x16, h0, a1, h3, a4, w0h, w3h, w6h, g1, e1, g4, e4 = ctx.saved_tensors
        M, group, eps1, eps4, S = ctx.meta
        need = ctx.needs_input_grad
        rec = Recorder(x16.device)
        G = dy.shape[0]                                                    # rows of dy: M, or M / group with the patch mean
        sv = torch.full((1,), S, dtype=torch.float32, device=x16.device)
        dy16 = rec.cast_to_f16(rec.affine_rows(dy.detach().float().contiguous().view(1, -1), sv).view(G, -1))
        kw = dict(dy_group=group, dy_skip=1, dy_scale=1.0 / (group - 1)) if group > 1 else {}
        da4 = rec.gemm(dy16, w6h.t().contiguous())                         # [G, 1024]
        dh3, dgb4 = rec.layernorm_backward(h3, da4, g4, e4, eps=eps4, act=ACT_LEAKY_RELU, **kw)
        a4m = rec.rows_mean(a4[1:], groups=G, count=group - 1, group_rows=group) if group > 1 else a4
        pend = {"w6": rec.wgrad(dy16, a4m), "b6": rec.colsum(dy16), "g4": dgb4[0], "e4": dgb4[1],
                "w3": rec.wgrad(dh3, a1), "b3": rec.colsum(dh3)}
        da1 = rec.gemm(dh3, w3h.t().contiguous())
        dh0, dgb1 = rec.layernorm_backward(h0, da1, g1, e1, eps=eps1, act=ACT_LEAKY_RELU)
        pend.update({"g1": dgb1[0], "e1": dgb1[1], "w0": rec.wgrad(dh0, x16), "b0": rec.colsum(dh0)})
        if need[0]:
            pend["x"] = rec.gemm(dh0, w0h.t().contiguous(), out_f32=True)
        inv = torch.full((1,), 1.0 / S, dtype=torch.float32, device=x16.device)
        outs = {k: rec.affine_rows(t.reshape(1, -1), inv).view(t.shape) for k, t in pend.items()}
        rec.run()
        order = ("x", "w0", "b0", "g1", "e1", "w3", "b3", "g4", "e4", "w6", "b6")
        return tuple(outs.get(k) if need[i] else None for i, k in enumerate(order)) + (None, None, None, None)


class PhotoVerseAdapter(nn.Module):
    def __init__(self, clip_embedding_dim=1024, cross_attention_dim=768, num_tokens=5):
        super().__init__()
        self.num_tokens = num_tokens
        for i in range(num_tokens):
            setattr(self, f"mapping_{i}", _mapping(clip_embedding_dim, cross_attention_dim))
            setattr(self, f"mapping_patch_{i}", _mapping(clip_embedding_dim, cross_attention_dim))

    @staticmethod
    def _mlp(rec: Recorder, seq: nn.Sequential, x, out=None):
        w0, b0 = _w(seq[0])
        h = rec.gemm(x, w0, bias=b0)
        h = rec.layernorm(h, seq[1].weight.detach().float().contiguous(), seq[1].bias.detach().float().contiguous(), eps=seq[1].eps,
                          act=ACT_LEAKY_RELU)
        w3, b3 = _w(seq[3])
        h = rec.gemm(h, w3, bias=b3)
        h = rec.layernorm(h, seq[4].weight.detach().float().contiguous(), seq[4].bias.detach().float().contiguous(), eps=seq[4].eps,
                          act=ACT_LEAKY_RELU)
        w6, b6 = _w(seq[6])
        return rec.gemm(h, w6, bias=b6, out=out)

    def _one(self, rec: Recorder, i: int, emb: torch.Tensor, out_rows: torch.Tensor):
        """emb (B, T, D) fp16 contiguous; writes (B, 768) into ``out_rows`` (a strided row view)."""
        B, T, D = emb.shape
        flat = emb.view(B * T, D)
        cls_rows = emb.view(B, T * D)[:, :D]                                   # emb[:, :1] as a strided row view
        self._mlp(rec, getattr(self, f"mapping_{i}"), cls_rows, out=out_rows)   # adapters.py:35 / :40
        pm = self._mlp(rec, getattr(self, f"mapping_patch_{i}"), flat)          # all T rows; the CLS row is simply not averaged
        rec.rows_mean(pm[1:], groups=B, count=T - 1, group_rows=T, out=out_rows, accumulate=True)   # .mean(dim=1) over emb[:, 1:]

    def _one_grad(self, i: int, emb: torch.Tensor) -> torch.Tensor:
        """Grad-mode token i: the two mapping MLPs as autograd nodes with HIP backward -> (B, 1, cout) fp32."""
        B, T, D = emb.shape
        e16 = emb.detach().to(torch.float16).contiguous() if not emb.requires_grad else emb.contiguous()
        gs = float(getattr(self, "grad_scale", 1.0))

        def run(seq, x, group):
            return _AdapterMLPFn.apply(x, seq[0].weight, seq[0].bias, seq[1].weight, seq[1].bias, seq[3].weight, seq[3].bias, seq[4].weight,
                                       seq[4].bias, seq[6].weight, seq[6].bias, seq[1].eps, seq[4].eps, group, gs)
        cls_out = run(getattr(self, f"mapping_{i}"), e16.view(B, T * D)[:, :D], 1)
        patch_out = run(getattr(self, f"mapping_patch_{i}"), e16.view(B * T, D), T)
        return _AddFn.apply(cls_out, patch_out).view(B, 1, -1)

    def forward(self, embs, token_index=None):
        require_cuda(embs[0], "image embeddings")
        dev = embs[0].device
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # training: per-token autograd nodes (adapters.py:30-44 semantics, HIP forward + backward)
            if token_index is not None and token_index != "full":
                return self._one_grad(int(token_index), embs[int(token_index)])
            return _CatFn.apply(*[self._one_grad(i, e) for i, e in enumerate(embs)])
        rec = Recorder(dev)
        cout = getattr(self, "mapping_0")[6].out_features

        def prep(e):
            e = e.contiguous()
            return rec.hold(e) if e.dtype == torch.float16 else rec.cast_to_f16(rec.hold(e.float())).view(e.shape)

        if token_index is not None and token_index != "full":                  # adapters.py:32-37
            token_index = int(token_index)
            emb = prep(embs[token_index])
            out = rec.empty((emb.shape[0], cout))
            self._one(rec, token_index, emb, out)
            rec.run()
            return out.view(emb.shape[0], 1, cout)
        B = embs[0].shape[0]
        n = len(embs)
        out = rec.empty((B, n * cout))                                          # torch.cat(..., dim=1) written in place (:39-44)
        for i, e in enumerate(embs):
            self._one(rec, i, prep(e), out[:, i * cout:(i + 1) * cout])
        rec.run()
        return out.view(B, n, cout)
