"""CLIP ViT-L/14 vision encoder and the PhotoVerse-patched CLIP text encoder on HIP kernels.

* ``CLIPVisionModel``: what the reference loads with ``CLIPVisionModel.from_pretrained("openai/clip-vit-large-patch14")``
  (``/root/reference/models/modeling_utils.py:59``) and calls as ``image_encoder(pixel_values_clip, output_hidden_states=True)``
  (``/root/reference/models/infer.py:76-78``).  Result indexing follows transformers: ``[0]`` last_hidden_state (no
  post-layernorm), ``[1]`` pooled, ``[2]`` the 25 hidden states with ``[2][0] = pre_layrnorm(embeddings)`` (``infer.py:80-84``).
* ``CLIPTextModel``: takes the reference's dict input (``/root/reference/models/clip.py:29-102``): token embedding ->
  concept injection (``clip.py:17-24``) -> + positions -> causal encoder -> final LN.  The injection is fused into the
  embedding kernel (``pv_clip_text_embed``).  ``patch_clip_text_transformer`` (``clip.py:115-119``) is kept as an API no-op:
  this class already has the patched forward.

Parameter names follow transformers 4.40 (``vision_model.*`` / ``text_model.*``) so HF checkpoints load.  The modules
only hold parameters; execution is a launch plan per input shape.  No CPU path.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from .ops import ACT_QUICK_GELU, Recorder, require_cuda


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError(f"{type(self).__name__} only holds parameters")


class _Attn(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (nn.Linear(dim, dim) for _ in range(4))


class _MLP(_Holder):
    def __init__(self, dim, inner):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(dim, inner), nn.Linear(inner, dim)


class _Layer(_Holder):
    def __init__(self, dim, inner):
        super().__init__()
        self.self_attn = _Attn(dim)
        self.layer_norm1 = nn.LayerNorm(dim)
        self.mlp = _MLP(dim, inner)
        self.layer_norm2 = nn.LayerNorm(dim)


class _Encoder(_Holder):
    def __init__(self, dim, inner, layers):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(dim, inner) for _ in range(layers)])


def _f16(t):
    return t.detach().to(torch.float16).contiguous()


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _encoder_layer(rec: Recorder, x, lyr: _Layer, batch, n, heads, causal):
    C = x.shape[1]
    sa = lyr.self_attn
    h = rec.layernorm(x, _f32(lyr.layer_norm1.weight), _f32(lyr.layer_norm1.bias), eps=lyr.layer_norm1.eps)
    wqkv = torch.cat([_f16(sa.q_proj.weight), _f16(sa.k_proj.weight), _f16(sa.v_proj.weight)], 0).contiguous()
    bqkv = torch.cat([_f32(sa.q_proj.bias), _f32(sa.k_proj.bias), _f32(sa.v_proj.bias)], 0).contiguous()
    qkv = rec.gemm(h, wqkv, bias=bqkv, rows_per_image=n)
    a = rec.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], batch=batch, heads=heads, nq=n, nk=n, d=C // heads, causal=causal)
    x = rec.gemm(a, _f16(sa.out_proj.weight), bias=_f32(sa.out_proj.bias), residual=x, rows_per_image=n)
    h = rec.layernorm(x, _f32(lyr.layer_norm2.weight), _f32(lyr.layer_norm2.bias), eps=lyr.layer_norm2.eps)
    f = rec.gemm(h, _f16(lyr.mlp.fc1.weight), bias=_f32(lyr.mlp.fc1.bias), act=ACT_QUICK_GELU, rows_per_image=n)
    return rec.gemm(f, _f16(lyr.mlp.fc2.weight), bias=_f32(lyr.mlp.fc2.bias), residual=x, rows_per_image=n)


class _CachedPlans(nn.Module):
    def __init__(self):
        super().__init__()
        self._plans: Dict[tuple, object] = {}

    def repack(self):
        self._plans.clear()

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.repack()
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._plans = {}
        return r


# ------------------------------------------------------------------------------------------------ vision
class _VisionEmbeddings(_Holder):
    def __init__(self, dim, image_size, patch):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(dim))
        self.patch_embedding = nn.Conv2d(3, dim, patch, stride=patch, bias=False)
        self.position_embedding = nn.Embedding((image_size // patch) ** 2 + 1, dim)


class _VisionTransformer(_Holder):
    def __init__(self, dim, inner, layers, image_size, patch):
        super().__init__()
        self.embeddings = _VisionEmbeddings(dim, image_size, patch)
        self.pre_layrnorm = nn.LayerNorm(dim)    # sic
        self.encoder = _Encoder(dim, inner, layers)
        self.post_layernorm = nn.LayerNorm(dim)


class _VisionOutput(tuple):
    """(last_hidden_state, pooler_output, hidden_states) with attribute access, like transformers' ModelOutput."""
    last_hidden_state = property(lambda s: s[0])
    pooler_output = property(lambda s: s[1])
    hidden_states = property(lambda s: s[2])


class CLIPVisionModel(_CachedPlans):
    def __init__(self, hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=24, image_size=224,
                 patch_size=14):
        super().__init__()
        self.config = SimpleNamespace(hidden_size=hidden_size, num_attention_heads=num_attention_heads, intermediate_size=intermediate_size,
                                      num_hidden_layers=num_hidden_layers, image_size=image_size, patch_size=patch_size)
        self.vision_model = _VisionTransformer(hidden_size, intermediate_size, num_hidden_layers, image_size, patch_size)

    def _plan(self, batch, dev):
        cfg, vm = self.config, self.vision_model
        rec = Recorder(dev)
        g = cfg.image_size // cfg.patch_size
        ntok, dim = g * g + 1, cfg.hidden_size
        kreal = 3 * cfg.patch_size ** 2
        kpad = (kreal + 63) // 64 * 64
        pixels = rec.empty((batch, 3, cfg.image_size, cfg.image_size), torch.float32)
        rows = rec.patchify(pixels, batch=batch, ch=3, img=cfg.image_size, patch=cfg.patch_size, kpad=kpad)
        wp = torch.zeros(dim, kpad, dtype=torch.float16, device=dev)
        wp[:, :kreal] = vm.embeddings.patch_embedding.weight.detach().reshape(dim, kreal).to(torch.float16)
        patches = rec.gemm(rows, wp, rows_per_image=g * g)
        x = rec.clip_vision_embed(patches, _f32(vm.embeddings.class_embedding), _f32(vm.embeddings.position_embedding.weight),
                                  batch=batch, ntok=ntok, dim=dim)
        x = rec.layernorm(x, _f32(vm.pre_layrnorm.weight), _f32(vm.pre_layrnorm.bias), eps=vm.pre_layrnorm.eps)
        hs = [x]
        for lyr in vm.encoder.layers:
            x = _encoder_layer(rec, x, lyr, batch, ntok, cfg.num_attention_heads, causal=False)
            hs.append(x)
        cls_rows = x.view(batch, ntok * dim)[:, :dim]          # strided row view of the CLS tokens
        pooled = rec.layernorm(cls_rows, _f32(vm.post_layernorm.weight), _f32(vm.post_layernorm.bias), eps=vm.post_layernorm.eps)
        return SimpleNamespace(rec=rec, pixels=pixels, hs=hs, pooled=pooled, ntok=ntok, dim=dim)

    def forward(self, pixel_values, output_hidden_states=True):
        require_cuda(pixel_values, "pixel_values")
        B = pixel_values.shape[0]
        key = (B, pixel_values.device)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._plan(B, pixel_values.device)
        plan.pixels.copy_(pixel_values)
        plan.rec.run()
        hs = tuple(h.view(B, plan.ntok, plan.dim).clone() for h in plan.hs)
        return _VisionOutput((hs[-1], plan.pooled.clone(), hs))


# ------------------------------------------------------------------------------------------------ text
class _TextEmbeddings(_Holder):
    def __init__(self, vocab, dim, max_pos):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, dim)
        self.position_embedding = nn.Embedding(max_pos, dim)


class _TextTransformer(_Holder):
    def __init__(self, vocab, dim, inner, layers, max_pos):
        super().__init__()
        self.embeddings = _TextEmbeddings(vocab, dim, max_pos)
        self.encoder = _Encoder(dim, inner, layers)
        self.final_layer_norm = nn.LayerNorm(dim)


class CLIPTextModel(_CachedPlans):
    def __init__(self, vocab_size=49408, hidden_size=768, num_attention_heads=12, intermediate_size=3072, num_hidden_layers=12,
                 max_position_embeddings=77):
        super().__init__()
        self.config = SimpleNamespace(vocab_size=vocab_size, hidden_size=hidden_size, num_attention_heads=num_attention_heads,
                                      max_position_embeddings=max_position_embeddings)
        self.text_model = _TextTransformer(vocab_size, hidden_size, intermediate_size, num_hidden_layers, max_position_embeddings)

    def _plan(self, batch, seq, n_concept, dev):
        cfg, tm = self.config, self.text_model
        rec = Recorder(dev)
        dim = cfg.hidden_size
        ids = rec.hold(torch.zeros((batch, seq), dtype=torch.int64, device=dev))
        concept = rec.empty((batch * max(n_concept, 1), dim)) if n_concept else None
        pidx = rec.hold(torch.zeros((batch,), dtype=torch.int64, device=dev)) if n_concept else None
        x = rec.clip_text_embed(ids, _f32(tm.embeddings.token_embedding.weight), _f32(tm.embeddings.position_embedding.weight),
                                concept, pidx, n_concept=n_concept, batch=batch, seq=seq, dim=dim)
        for lyr in tm.encoder.layers:
            x = _encoder_layer(rec, x, lyr, batch, seq, cfg.num_attention_heads, causal=True)
        out = rec.layernorm(x, _f32(tm.final_layer_norm.weight), _f32(tm.final_layer_norm.bias), eps=tm.final_layer_norm.eps)
        return SimpleNamespace(rec=rec, ids=ids, concept=concept, pidx=pidx, out=out)

    def forward(self, input_ids=None):
        if input_ids is None:                                      # clip.py:47-48
            raise ValueError("You have to specify either input_ids")
        ids = input_ids["text_input_ids"]                          # clip.py:50-52
        concept = input_ids.get("concept_text_embeddings", None)
        pidx = input_ids.get("concept_placeholder_idx", None)
        require_cuda(ids, "text_input_ids")
        ids = ids.view(-1, ids.shape[-1])
        B, S = ids.shape
        if S > self.config.max_position_embeddings:
            raise IndexError(f"sequence length {S} exceeds max_position_embeddings {self.config.max_position_embeddings}")
        lo, hi = int(ids.min()), int(ids.max())       # one host sync, pre-loop; nn.Embedding would raise / device-assert here
        if lo < 0 or hi >= self.config.vocab_size:
            raise IndexError(f"token id out of range [0, {self.config.vocab_size})")
        E = 0 if concept is None else concept.shape[1]
        if E:
            # the reference's slice assignment (clip.py:17-24) raises when the concept rows do not fit the sequence; the fused
            # embed kernel would read out of bounds (negative index) or truncate silently - check on the host like the ids
            plo, phi = int(pidx.min()), int(pidx.max())
            if plo < 0 or phi + E > S:
                raise IndexError(f"concept_placeholder_idx must satisfy 0 <= idx and idx + {E} <= {S} (got [{plo}, {phi}])")
        key = (B, S, E, ids.device)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._plans[key] = self._plan(B, S, E, ids.device)
        plan.ids.copy_(ids)
        if E:
            plan.concept.copy_(concept.reshape(B * E, -1))
            plan.pidx.copy_(pidx.reshape(-1))
        plan.rec.run()
        last = plan.out.view(B, S, -1).clone()
        pooled = last[torch.arange(B, device=ids.device), ids.to(torch.int).argmax(dim=-1)]     # clip.py:90-92 (indexing only)
        return (last, pooled)


def patch_clip_text_transformer(text_encoder):
    """API parity with ``/root/reference/models/clip.py:115-119``.  ``photoverse_amd.CLIPTextModel`` already implements the
    dict-input forward, so there is nothing to swap; other encoders are rejected loudly instead of silently no-op'ing
    (which is what the reference's by-class-name patch would do on a class it does not know)."""
    if not isinstance(text_encoder, CLIPTextModel):
        raise TypeError("patch_clip_text_transformer expects photoverse_amd.clip.CLIPTextModel")
    return text_encoder
