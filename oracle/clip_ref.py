"""Oracle: CLIP ViT-L/14 vision encoder and the PhotoVerse-patched CLIP text
forward.  TEST INFRASTRUCTURE.

* Vision: [EXT] ``transformers`` ``CLIPVisionModel("openai/clip-vit-large-patch14")``
  loaded at ``/root/reference/models/modeling_utils.py:59`` and called with
  ``output_hidden_states=True`` at ``/root/reference/models/infer.py:76-78``;
  consumers index ``[0]`` (last_hidden_state, NO post-layernorm) and ``[2][i]``
  (25 hidden states, ``[2][0]`` = pre_layrnorm(embeddings)) at ``infer.py:80-84``.
* Text: ``clip_text_transformer_forward`` (``/root/reference/models/clip.py:29-102``)
  with ``_inject_concept_embeddings`` (``clip.py:17-24``).

PINNED (version-skewed): ``tests/test_oracle_pins.py`` compares both against the
installed transformers 5.x models with shared random weights; injection is
pinned on the worked example in ``clip.py:21-23``.  Parameter names follow
transformers 4.40 (``vision_model.*`` / ``text_model.*``).
"""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.heads = heads
        self.q_proj = nn.Linear(dim, dim)
        self.k_proj = nn.Linear(dim, dim)
        self.v_proj = nn.Linear(dim, dim)
        self.out_proj = nn.Linear(dim, dim)

    def forward(self, x, causal=False):
        b, n, c = x.shape
        d = c // self.heads
        q = self.q_proj(x).view(b, n, self.heads, d).transpose(1, 2)
        k = self.k_proj(x).view(b, n, self.heads, d).transpose(1, 2)
        v = self.v_proj(x).view(b, n, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, is_causal=causal)
        return self.out_proj(o.transpose(1, 2).reshape(b, n, c))


class _MLP(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.fc1 = nn.Linear(dim, inner)
        self.fc2 = nn.Linear(inner, dim)

    def forward(self, x):
        return self.fc2(quick_gelu(self.fc1(x)))


class _Layer(nn.Module):
    def __init__(self, dim, heads, inner):
        super().__init__()
        self.self_attn = _Attn(dim, heads)
        self.layer_norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.mlp = _MLP(dim, inner)
        self.layer_norm2 = nn.LayerNorm(dim, eps=1e-5)

    def forward(self, x, causal=False):
        x = x + self.self_attn(self.layer_norm1(x), causal)
        return x + self.mlp(self.layer_norm2(x))


class _Encoder(nn.Module):
    def __init__(self, dim, heads, inner, layers):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(dim, heads, inner) for _ in range(layers)])


class _VisionEmbeddings(nn.Module):
    def __init__(self, dim, image_size, patch):
        super().__init__()
        self.class_embedding = nn.Parameter(torch.randn(dim))
        self.patch_embedding = nn.Conv2d(3, dim, patch, stride=patch, bias=False)
        self.position_embedding = nn.Embedding((image_size // patch) ** 2 + 1, dim)

    def forward(self, pixel_values):
        b = pixel_values.shape[0]
        p = self.patch_embedding(pixel_values).flatten(2).transpose(1, 2)
        x = torch.cat([self.class_embedding.expand(b, 1, -1), p], dim=1)
        return x + self.position_embedding.weight[None]


class _VisionTransformer(nn.Module):
    def __init__(self, dim, heads, inner, layers, image_size, patch):
        super().__init__()
        self.embeddings = _VisionEmbeddings(dim, image_size, patch)
        self.pre_layrnorm = nn.LayerNorm(dim, eps=1e-5)   # sic: transformers' spelling
        self.encoder = _Encoder(dim, heads, inner, layers)
        self.post_layernorm = nn.LayerNorm(dim, eps=1e-5)


class CLIPVisionModelRef(nn.Module):
    def __init__(self, hidden_size=1024, num_attention_heads=16, intermediate_size=4096, num_hidden_layers=24,
                 image_size=224, patch_size=14):
        super().__init__()
        self.config = SimpleNamespace(hidden_size=hidden_size, num_hidden_layers=num_hidden_layers, image_size=image_size,
                                      patch_size=patch_size, num_attention_heads=num_attention_heads,
                                      intermediate_size=intermediate_size)
        self.vision_model = _VisionTransformer(hidden_size, num_attention_heads, intermediate_size, num_hidden_layers,
                                               image_size, patch_size)

    def forward(self, pixel_values, output_hidden_states=True):
        vm = self.vision_model
        x = vm.pre_layrnorm(vm.embeddings(pixel_values))
        hs = (x,)
        for layer in vm.encoder.layers:
            x = layer(x)
            hs += (x,)
        pooled = vm.post_layernorm(x[:, 0])
        return (x, pooled, hs)


# ---------------------------------------------------------------------------
def inject_concept_embeddings_ref(inputs_embeds, concept_text_embeddings, concept_placeholder_idx):
    """``/root/reference/models/clip.py:17-24``: per row, the tokens after the
    placeholder are shifted right by ``emb_length - 1`` (tail truncated) and
    ``[idx, idx+emb_length)`` is overwritten with the concept embeddings."""
    new = inputs_embeds.clone()
    n = concept_text_embeddings.shape[1]
    for b, idx in enumerate(concept_placeholder_idx):
        idx = int(idx)
        left = new.shape[1] - n - idx
        new[b, idx + n:] = inputs_embeds[b, idx + 1: idx + 1 + left]
        new[b, idx: idx + n] = concept_text_embeddings[b]
    return new


class _TextEmbeddings(nn.Module):
    def __init__(self, vocab, dim, max_pos):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, dim)
        self.position_embedding = nn.Embedding(max_pos, dim)


class _TextTransformer(nn.Module):
    def __init__(self, vocab, dim, heads, inner, layers, max_pos):
        super().__init__()
        self.embeddings = _TextEmbeddings(vocab, dim, max_pos)
        self.encoder = _Encoder(dim, heads, inner, layers)
        self.final_layer_norm = nn.LayerNorm(dim, eps=1e-5)


class CLIPTextModelRef(nn.Module):
    """SD-v1.5 text encoder with the reference's dict-input forward already
    installed (``patch_clip_text_transformer``, ``clip.py:115-119``)."""

    def __init__(self, vocab_size=49408, hidden_size=768, num_attention_heads=12, intermediate_size=3072,
                 num_hidden_layers=12, max_position_embeddings=77):
        super().__init__()
        self.config = SimpleNamespace(vocab_size=vocab_size, hidden_size=hidden_size)
        self.text_model = _TextTransformer(vocab_size, hidden_size, num_attention_heads, intermediate_size,
                                           num_hidden_layers, max_position_embeddings)

    def forward(self, input_ids):
        if input_ids is None:                                    # clip.py:47-48
            raise ValueError("You have to specify either input_ids")
        tm = self.text_model
        ids = input_ids["text_input_ids"]                        # clip.py:50-52
        concept = input_ids.get("concept_text_embeddings", None)
        idx = input_ids.get("concept_placeholder_idx", None)
        ids = ids.view(-1, ids.shape[-1])
        emb = tm.embeddings.token_embedding(ids)                 # clip.py:57
        if concept is not None:
            emb = inject_concept_embeddings_ref(emb, concept, idx)   # clip.py:58-59
        x = emb + tm.embeddings.position_embedding.weight[None, : ids.shape[1]]   # clip.py:63
        for layer in tm.encoder.layers:                          # clip.py:75-82, causal mask :67-69
            x = layer(x, causal=True)
        x = tm.final_layer_norm(x)                               # clip.py:84-85
        pooled = x[torch.arange(x.shape[0]), ids.to(torch.int).argmax(dim=-1)]   # clip.py:90-92
        return (x, pooled)
