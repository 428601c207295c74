"""CPU oracle of the pre-tokenisation step (reference trainer.py:136-214) -- TEST INFRASTRUCTURE ONLY, like the rest of
oracle/: imported by tests/, __graft_entry__.smoke() and bench.py's CPU-baseline legs, never by the product package.

The reference's algorithm for this step IS a call into its third-party dependency `regex` (uv.lock pins 2025.11.3; this
image has the version printed by `regex.__version__`): decode a chunk as UTF-8 (:156-161), `regex.findall` with the GPT-2
split pattern (:163) preceded by the escaped special tokens in configuration order (:165-167), keep the non-empty matches
as UTF-8 byte strings (:169-170).  Chunks are independent texts (:172-198).  Pinned by tests/golden/g6_pretokens.json
(hashes of the reference's own `_preprocess_corpus` outputs, chunked variants included)."""
from __future__ import annotations

import regex

GPT2_SPLIT = r"""'(?:[sdmt]|ll|ve|re)| ?\p{L}+| ?\p{N}+| ?[^\s\p{L}\p{N}]+|\s+(?!\S)|\s+"""  # trainer.py:163


def split_pattern(special_tokens=()) -> "regex.Pattern[str]":
    pat = GPT2_SPLIT
    if special_tokens:  # trainer.py:165-167
        pat = "|".join(regex.escape(t) for t in special_tokens) + "|" + pat
    return regex.compile(pat)


def pretokenize(data: bytes, special_tokens=(), chunk_starts=(0,)) -> list[bytes]:
    """Pre-tokens of `data` as byte strings; every chunk [chunk_starts[k], chunk_starts[k+1]) is a text of its own.
    Raises UnicodeDecodeError exactly where the reference would (its .start is relative to the chunk)."""
    pat = split_pattern(special_tokens)
    out: list[bytes] = []
    bounds = list(chunk_starts) + [len(data)]
    for a, b in zip(bounds[:-1], bounds[1:]):
        out += [t.encode("utf-8") for t in pat.findall(data[a:b].decode("utf-8")) if t]  # trainer.py:156, 169-170
    return out
