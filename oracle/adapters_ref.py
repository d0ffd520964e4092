"""Oracle: restatement of ``PhotoVerseAdapter``
(``/root/reference/models/adapters.py:5-44``).  TEST INFRASTRUCTURE.

PINNED: ``oracle/make_golden.py`` imports the real reference module in the
build container and stores seeded input/output vectors in
``tests/golden/adapter_golden.pt``; ``tests/test_oracle_pins.py`` checks this
restatement against them.  State-dict names equal the reference's
(``mapping_{i}.{0,1,3,4,6}.*`` / ``mapping_patch_{i}.*``).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _mapping(cin, cout):
    # adapters.py:14-20 (identical for mapping_i and mapping_patch_i, :22-28)
    return nn.Sequential(nn.Linear(cin, 1024), nn.LayerNorm(1024), nn.LeakyReLU(),
                         nn.Linear(1024, 1024), nn.LayerNorm(1024), nn.LeakyReLU(),
                         nn.Linear(1024, cout))


class PhotoVerseAdapterRef(nn.Module):
    def __init__(self, clip_embedding_dim=1024, cross_attention_dim=768, num_tokens=5):
        super().__init__()
        self.num_tokens = num_tokens
        for i in range(num_tokens):  # registration order matters for seeded init: mapping_i then mapping_patch_i
            setattr(self, f"mapping_{i}", _mapping(clip_embedding_dim, cross_attention_dim))
            setattr(self, f"mapping_patch_{i}", _mapping(clip_embedding_dim, cross_attention_dim))

    def _one(self, i, emb):
        # adapters.py:35-36 / :40-41: CLS token through mapping_i, mean over the 256 patch tokens of mapping_patch_i
        return getattr(self, f"mapping_{i}")(emb[:, :1]) + getattr(self, f"mapping_patch_{i}")(emb[:, 1:]).mean(dim=1, keepdim=True)

    def forward(self, embs, token_index=None):
        if token_index is not None and token_index != "full":   # adapters.py:32-37
            token_index = int(token_index)
            return self._one(token_index, embs[token_index])
        return torch.cat([self._one(i, emb) for i, emb in enumerate(embs)], dim=1)  # :39-44
