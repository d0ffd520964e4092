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


def _mapping(cin, cout):
    return nn.Sequential(nn.Linear(cin, 1024), nn.LayerNorm(1024), nn.LeakyReLU(),
                         nn.Linear(1024, 1024), nn.LayerNorm(1024), nn.LeakyReLU(),
                         nn.Linear(1024, cout))


def _w(lin):
    return lin.weight.detach().to(torch.float16).contiguous(), lin.bias.detach().float().contiguous()


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

    def forward(self, embs, token_index=None):
        require_cuda(embs[0], "image embeddings")
        dev = embs[0].device
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
