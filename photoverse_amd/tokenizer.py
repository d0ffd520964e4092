"""Tokenizers for the prompt side of the path (``/root/reference/models/modeling_utils.py:55``,
``datasets/utils.py:160-199``, ``models/infer.py:43-49``).

* ``CLIPBPETokenizer`` - the CLIP byte-level BPE ([EXT] transformers ``CLIPTokenizer``, the no-ftfy branch), built from a LOCAL
  ``vocab.json`` + ``merges.txt`` (``<model dir>/tokenizer/``): lower-case + whitespace clean-up, the CLIP split pattern,
  byte -> unicode alphabet, greedy lowest-rank merges with the ``</w>`` end-of-word marker, ``<|startoftext|>`` /
  ``<|endoftext|>`` wrapping, truncation to ``model_max_length`` and ``<|endoftext|>`` padding.  Checked in
  ``tests/test_host_cpu.py`` against the installed ``transformers`` tokenizer on a synthetic vocabulary.
* ``SyntheticCLIPTokenizer`` - stand-in when no vocabulary files exist (this image has no network and
  ``CLIPTokenizer.from_pretrained`` returns an empty shell offline, SURVEY.md 8c "tokenizer trap").  It only knows what the hot
  path needs: the ids of the EMPTY prompt of the unconditional branch - ``[BOS, EOS, EOS, ...]`` with CLIP's ids - and a
  deterministic hash-based id assignment for synthetic prompts (NOT BPE; plumbing tests and benchmarks only).
"""
from __future__ import annotations

import json
import os
from functools import lru_cache
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


@lru_cache()
def bytes_to_unicode():
    """The GPT-2 / CLIP byte alphabet: every byte maps to a printable unicode character (printable bytes to themselves, the
    others to code points from 256 up)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


class CLIPBPETokenizer:
    """Byte-level BPE with CLIP's conventions; same call signature / return as the slice of ``CLIPTokenizer`` the reference
    uses: ``tok(text, padding="max_length", truncation=True, max_length=tok.model_max_length, return_tensors="pt").input_ids``."""

    model_max_length = 77

    def __init__(self, vocab_file: str, merges_file: str, model_max_length: int = 77):
        import regex
        with open(vocab_file, encoding="utf-8") as fh:
            self.encoder = json.load(fh)
        with open(merges_file, encoding="utf-8") as fh:
            lines = fh.read().strip().split("\n")
        if lines and lines[0].startswith("#"):          # "#version: 0.2"
            lines = lines[1:]
        merges = [tuple(ln.split()) for ln in lines if ln.strip()]
        self.bpe_ranks = {m: i for i, m in enumerate(merges)}
        self.byte_encoder = bytes_to_unicode()
        self.bos_token, self.eos_token = "<|startoftext|>", "<|endoftext|>"
        self.bos_token_id, self.eos_token_id = self.encoder[self.bos_token], self.encoder[self.eos_token]
        self.pad_token_id = self.unk_token_id = self.eos_token_id
        self.vocab_size = len(self.encoder)
        self.model_max_length = model_max_length
        self.pat = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                                 regex.IGNORECASE)
        self._cache = {self.bos_token: self.bos_token, self.eos_token: self.eos_token}

    @classmethod
    def from_pretrained(cls, path, subfolder="tokenizer"):
        d = os.path.join(path, subfolder) if subfolder else path
        return cls(os.path.join(d, "vocab.json"), os.path.join(d, "merges.txt"))

    def bpe(self, token: str) -> str:
        if token in self._cache:
            return self._cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        while len(word) > 1:
            pairs = {(word[i], word[i + 1]) for i in range(len(word) - 1)}
            bigram = min(pairs, key=lambda pr: self.bpe_ranks.get(pr, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new, i = [], 0
            while i < len(word):
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    new.append(first + second)
                    i += 2
                else:
                    new.append(word[i])
                    i += 1
            word = tuple(new)
        out = " ".join(word)
        self._cache[token] = out
        return out

    def tokenize(self, text: str):
        text = " ".join(text.split()).strip().lower()
        out = []
        for tok in self.pat.findall(text):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            out.extend(self.bpe(tok).split(" "))
        return out

    def __call__(self, text, padding="max_length", max_length=None, truncation=True, return_tensors="pt"):
        if isinstance(text, str):
            text = [text]
        L = max_length or self.model_max_length
        rows = []
        for t in text:
            ids = [self.encoder.get(tk, self.unk_token_id) for tk in self.tokenize(t)]
            if truncation:
                ids = ids[: L - 2]
            ids = [self.bos_token_id] + ids + [self.eos_token_id]
            if padding == "max_length":
                ids = ids + [self.pad_token_id] * (L - len(ids))
            rows.append(ids)
        return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.int64))


def load_tokenizer(model_dir=None):
    """``CLIPTokenizer.from_pretrained(path, subfolder="tokenizer")`` of ``modeling_utils.py:55``: the real BPE when the
    vocabulary files are present under ``model_dir``, else the synthetic stand-in."""
    if model_dir is not None:
        d = os.path.join(str(model_dir), "tokenizer")
        if os.path.exists(os.path.join(d, "vocab.json")) and os.path.exists(os.path.join(d, "merges.txt")):
            return CLIPBPETokenizer.from_pretrained(str(model_dir))
    return SyntheticCLIPTokenizer()
