"""Stand-in for ``CLIPTokenizer`` when no vocabulary files are available (this image has no network and
``CLIPTokenizer.from_pretrained`` returns an empty shell offline, SURVEY.md 8c "tokenizer trap").

It only knows what the hot path needs: the ids of the EMPTY prompt that ``run_inference`` builds for the unconditional
branch (``/root/reference/models/infer.py:43-49``) - ``[BOS, EOS, EOS, ...]`` with CLIP's ids - and a deterministic
hash-based id assignment for synthetic prompts (NOT the real BPE; for plumbing tests and benchmarks only).
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

BOS, EOS = 49406, 49407


class SyntheticCLIPTokenizer:
    model_max_length = 77
    vocab_size = 49408

    def __call__(self, text, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        if isinstance(text, str):
            text = [text]
        L = max_length or self.model_max_length
        rows = []
        for t in text:
            words = t.split()
            ids = [BOS] + [1000 + (hash_word(w) % 40000) for w in words][: L - 2] + [EOS]
            ids = ids + [EOS] * (L - len(ids))
            rows.append(ids)
        return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.int64))


def hash_word(w: str) -> int:
    h = 2166136261
    for ch in w.encode("utf-8"):
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h
