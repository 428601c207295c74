"""Byte-level BPE tokenizer: consumer of the trainer's (vocab, merges).

Plain Python, same public API as the reference `src/yet_another_bpe/tokenizer.py:35-398`
(encode, decode, encode_batch, decode_batch, from_file, vocab_size, special_tokens, get_vocab,
clear_cache, cache_info, _encode_word).  Inference is outside the accelerated hot path;
this module exists so that models trained on the GPU can be used and round-tripped.
"""
from __future__ import annotations

import json
from collections.abc import Sequence
from functools import lru_cache
from pathlib import Path

import regex

_GPT2_SPLIT = r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""
_WORD_CACHE = 8192


class BBPETokenizer:
    """Applies learned merges, lowest rank first (leftmost on ties), to GPT-2 pre-tokens."""

    def __init__(self, vocab: dict[bytes, int] | None = None, merges: list[tuple[bytes, bytes]] | None = None,
                 special_tokens: list[str] | None = None) -> None:
        self._vocab: dict[bytes, int] = vocab or {}
        self._vocab_inv: dict[int, bytes] = {i: t for t, i in self._vocab.items()}
        self._merges: list[tuple[bytes, bytes]] = merges or []
        self._special_tokens: list[str] = special_tokens or []
        self._special_set = frozenset(self._special_tokens)
        self._rank: dict[tuple[bytes, bytes], int] = {pair: i for i, pair in enumerate(self._merges)}
        self._pattern = regex.compile(_GPT2_SPLIT)
        # specials are split out first, longest first (reference tokenizer.py:100-102)
        self._special_pattern = None
        if self._special_tokens:
            ordered = sorted(self._special_tokens, key=len, reverse=True)
            self._special_pattern = regex.compile("(" + "|".join(regex.escape(t) for t in ordered) + ")")
        self._word_ids = lru_cache(maxsize=_WORD_CACHE)(self._word_ids_uncached)

    # ------------------------------------------------------------------ persistence (tokenizer.py:106-150)
    @classmethod
    def from_file(cls, model_dir: str | Path) -> "BBPETokenizer":
        d = Path(model_dir)
        with open(d / "vocab.json", encoding="utf-8") as f:
            vocab = {k.encode("latin-1"): v for k, v in json.load(f).items()}
        merges: list[tuple[bytes, bytes]] = []
        with open(d / "merges.txt", encoding="utf-8") as f:
            for line in f:
                line = line.rstrip("\n")
                if not line:
                    continue
                left, sep, right = line.partition(" ")  # first space splits (tokenizer.py:137)
                if sep:
                    merges.append((left.encode("latin-1"), right.encode("latin-1")))
        specials: list[str] = []
        sp = d / "special_tokens.json"
        if sp.exists():
            with open(sp, encoding="utf-8") as f:
                specials = list(json.load(f))
        return cls(vocab=vocab, merges=merges, special_tokens=specials)

    @classmethod
    def from_file_lossless(cls, model_dir: str | Path) -> "BBPETokenizer":
        """Loads what BBPETrainer.save_lossless wrote (hex files): every token and merge exactly as trained."""
        d = Path(model_dir)
        with open(d / "vocab.hex.json", encoding="ascii") as f:
            vocab = {bytes.fromhex(k): v for k, v in json.load(f).items()}
        merges: list[tuple[bytes, bytes]] = []
        with open(d / "merges.hex", encoding="ascii") as f:
            for line in f:
                left, _, right = line.strip().partition(" ")
                if left or right:
                    merges.append((bytes.fromhex(left), bytes.fromhex(right)))
        specials: list[str] = []
        if (d / "special_tokens.json").exists():
            with open(d / "special_tokens.json", encoding="utf-8") as f:
                specials = list(json.load(f))
        return cls(vocab=vocab, merges=merges, special_tokens=specials)

    # ------------------------------------------------------------------ encode
    def _word_ids_uncached(self, word: str) -> tuple[int, ...]:
        data = word.encode("utf-8")
        if not data:
            return ()
        unk = self._vocab.get(b"[UNK]", 0)
        parts = [bytes([b]) for b in data]
        rank = self._rank
        while len(parts) > 1:
            best_rank, best_i = None, -1
            for i in range(len(parts) - 1):
                r = rank.get((parts[i], parts[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best_rank, best_i = r, i
            if best_rank is None:
                break
            parts[best_i:best_i + 2] = [parts[best_i] + parts[best_i + 1]]
        return tuple(self._vocab.get(p, unk) for p in parts)

    def _encode_word(self, word: str) -> list[int]:
        return list(self._word_ids(word))

    def _encode_plain(self, text: str, out: list[int]) -> None:
        for pre in self._pattern.findall(text):
            out.extend(self._word_ids(pre))

    def encode(self, text: str) -> list[int]:
        if not text:
            return []
        ids: list[int] = []
        if self._special_pattern is None:
            self._encode_plain(text, ids)
            return ids
        for part in self._special_pattern.split(text):
            if not part:
                continue
            if part in self._special_set:
                tid = self._vocab.get(part.encode("utf-8"))
                if tid is not None:
                    ids.append(tid)
            else:
                self._encode_plain(part, ids)
        return ids

    def encode_batch(self, texts: Sequence[str]) -> list[list[int]]:
        return [self.encode(t) for t in texts]

    # ------------------------------------------------------------------ decode (tokenizer.py:324-349)
    def decode(self, ids: Sequence[int]) -> str:
        if not ids:
            return ""
        inv = self._vocab_inv
        data = b"".join(inv[i] for i in ids if i in inv)  # unknown ids are skipped
        try:
            return data.decode("utf-8")
        except UnicodeDecodeError:
            return data.decode("utf-8", errors="replace")

    def decode_batch(self, ids_batch: Sequence[Sequence[int]]) -> list[str]:
        return [self.decode(ids) for ids in ids_batch]

    # ------------------------------------------------------------------ introspection
    @property
    def vocab_size(self) -> int:
        return len(self._vocab)

    @property
    def special_tokens(self) -> list[str]:
        return list(self._special_tokens)

    def get_vocab(self) -> dict[str, int]:
        return {t.decode("latin-1"): i for t, i in self._vocab.items()}

    def clear_cache(self) -> None:
        self._word_ids.cache_clear()

    def cache_info(self) -> str:
        info = self._word_ids.cache_info()
        return f"hits={info.hits}, misses={info.misses}, size={info.currsize}/{info.maxsize}"
