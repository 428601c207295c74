"""Shared test helpers (golden loaders, base vocab, flattening)."""
from __future__ import annotations

import json
from collections import Counter
from functools import lru_cache
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def base_tokens(special_tokens) -> list[bytes]:
    """_init_base_vocab of the reference (trainer.py:119-134) as an id-ordered list."""
    toks = [bytes([b]) for b in range(256)]
    seen = set(toks)
    for s in special_tokens:
        tb = s.encode("utf-8") if isinstance(s, str) else bytes(s)
        if tb not in seen:
            seen.add(tb)
            toks.append(tb)
    return toks


def flatten(words) -> tuple[np.ndarray, np.ndarray]:
    lens = np.fromiter((len(w) for w in words), dtype=np.uint64, count=len(words))
    off = np.zeros(len(words) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    flat = np.frombuffer(b"".join(bytes(w) for w in words), dtype=np.uint8).copy()
    return flat, off


def pooled(words) -> tuple[list[bytes], np.ndarray]:
    c = Counter(bytes(w) for w in words)
    return list(c), np.fromiter(c.values(), dtype=np.uint64, count=len(c))


def read_hex_merges(path: Path) -> list[tuple[bytes, bytes]]:
    out = []
    for line in path.read_text().splitlines():
        a, b = line.split(" ")
        out.append((bytes.fromhex(a), bytes.fromhex(b)))
    return out


@lru_cache(maxsize=None)
def golden_cases() -> list[dict]:
    cases = json.loads((GOLDEN / "g345_cases.json").read_text())
    for c in cases:
        c["words_b"] = [bytes.fromhex(w) for w in c["words"]]
        c["merges_b"] = [(bytes.fromhex(a), bytes.fromhex(b)) for a, b in c["merges"]]
        c["vocab_b"] = {bytes.fromhex(k): v for k, v in c["vocab"].items()}
    return cases


@lru_cache(maxsize=None)
def corpus_en_words() -> list[bytes]:
    """Pre-tokens of corpus.en (specials ["<|endoftext|>"], one 1 GiB chunk) via the product's host pre-tokenizer,
    which tests/test_host_api.py pins against the reference (G6)."""
    from yet_another_bpe.trainer import BBPETrainer, BBPETrainerConfig

    t = BBPETrainer(BBPETrainerConfig(max_workers=1, chunk_size_bytes=1 << 30, special_tokens=["<|endoftext|>"]))
    return [bytes(s) for s in t._preprocess_corpus([GOLDEN / "corpus.en"])]


def gpt2_bytes_to_unicode() -> dict[int, str]:
    """GPT-2 printable byte encoding (needed to read the reference's merges fixture, tests/common.py:9-54)."""
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, (chr(c) for c in cs)))


def read_gpt2_merges(path: Path) -> list[tuple[bytes, bytes]]:
    dec = {v: k for k, v in gpt2_bytes_to_unicode().items()}
    out = []
    for line in path.read_text(encoding="utf-8").splitlines():
        a, b = line.rstrip().split(" ")
        out.append((bytes(dec[ch] for ch in a), bytes(dec[ch] for ch in b)))
    return out
