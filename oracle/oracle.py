"""ctypes front-end of the CPU oracle (oracle/bpe_oracle.c) + a tiny pure-Python recount restatement.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (yet-another-bpe_amd/) never imports this module.

Reference being restated: src/yet_another_bpe/trainer.py:119-134 (_init_base_vocab) and
:216-302 (_merge_loop) of DreamOneX/yet-another-bpe.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from collections import Counter
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    so = _HERE / "libbpe_oracle.so"
    src = _HERE / "bpe_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "libbpe_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(str(build()))
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        lib.bpe_oracle_train.restype = vp
        lib.bpe_oracle_train.argtypes = [vp, vp, u64, vp, vp, u32, u64, u64, ctypes.c_double]
        lib.bpe_oracle_train_weighted.restype = vp
        lib.bpe_oracle_train_weighted.argtypes = [vp, vp, vp, u64, vp, vp, u32, u64, u64, ctypes.c_double]
        for name in ("bpe_oracle_n_merges", "bpe_oracle_n_tokens"):
            getattr(lib, name).restype = u32
            getattr(lib, name).argtypes = [vp]
        for name in ("bpe_oracle_n_unique_words", "bpe_oracle_n_pairs_initial"):
            getattr(lib, name).restype = u64
            getattr(lib, name).argtypes = [vp]
        lib.bpe_oracle_get_merges.restype = None
        lib.bpe_oracle_get_merges.argtypes = [vp, vp, vp, vp, vp]
        lib.bpe_oracle_token_len.restype = u32
        lib.bpe_oracle_token_len.argtypes = [vp, u32]
        lib.bpe_oracle_token_bytes.restype = None
        lib.bpe_oracle_token_bytes.argtypes = [vp, u32, vp]
        lib.bpe_oracle_free.restype = None
        lib.bpe_oracle_free.argtypes = [vp]
        _LIB = lib
    return _LIB


def flatten(sequences) -> tuple[np.ndarray, np.ndarray]:
    """list[list[int]] (or list[bytes]) -> (flat u8 bytes, u64 offsets of length n+1)."""
    lens = np.fromiter((len(s) for s in sequences), dtype=np.uint64, count=len(sequences))
    off = np.zeros(len(sequences) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    flat = np.frombuffer(b"".join(bytes(s) for s in sequences), dtype=np.uint8)
    return flat.copy() if flat.size else np.zeros(0, dtype=np.uint8), off


def train_flat(flat: np.ndarray, off: np.ndarray, vocab_size: int, min_frequency: int, special_tokens,
               return_ids: bool = False, max_seconds: float = 0.0, freq: np.ndarray | None = None):
    """Run the C oracle on flat words.  Returns (vocab: dict[bytes,int], merges: list[(bytes,bytes)])
    exactly as BBPETrainer._merge_loop does (trainer.py:302); with return_ids also the id triples/counts.
    freq: occurrences per input word (words pooled by the caller, trainer.py:221-225)."""
    lib = _lib()
    flat = np.ascontiguousarray(flat, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    fq = None if freq is None else np.ascontiguousarray(freq, dtype=np.uint64)
    n_words = len(off) - 1
    sp = [t.encode("utf-8") if isinstance(t, str) else bytes(t) for t in special_tokens]
    sp_bytes = np.frombuffer(b"".join(sp) or b"\0", dtype=np.uint8).copy()
    sp_off = np.zeros(len(sp) + 1, dtype=np.uint32)
    if sp:
        sp_off[1:] = np.cumsum([len(t) for t in sp])
    h = lib.bpe_oracle_train_weighted(flat.ctypes.data if flat.size else None, off.ctypes.data, None if fq is None else fq.ctypes.data, n_words,
                                      sp_bytes.ctypes.data, sp_off.ctypes.data, len(sp), int(vocab_size), int(min_frequency), float(max_seconds))
    try:
        nm = lib.bpe_oracle_n_merges(h)
        nt = lib.bpe_oracle_n_tokens(h)
        left = np.zeros(nm, dtype=np.uint32)
        right = np.zeros(nm, dtype=np.uint32)
        merged = np.zeros(nm, dtype=np.uint32)
        count = np.zeros(nm, dtype=np.uint64)
        if nm:
            lib.bpe_oracle_get_merges(h, left.ctypes.data, right.ctypes.data, merged.ctypes.data, count.ctypes.data)
        toks = []
        for i in range(nt):
            n = lib.bpe_oracle_token_len(h, i)
            buf = (ctypes.c_uint8 * max(n, 1))()
            lib.bpe_oracle_token_bytes(h, i, buf)
            toks.append(bytes(buf[:n]))
        stats = {"unique_words": lib.bpe_oracle_n_unique_words(h), "pairs_initial": lib.bpe_oracle_n_pairs_initial(h)}
    finally:
        lib.bpe_oracle_free(h)
    vocab = {t: i for i, t in enumerate(toks)}
    merges = [(toks[int(l)], toks[int(r)]) for l, r in zip(left, right)]
    if return_ids:
        return vocab, merges, {"left": left, "right": right, "merged": merged, "count": count, **stats}
    return vocab, merges


def merge_loop(sequences, vocab_size: int, min_frequency: int, special_tokens):
    """Same signature shape as the reference's _merge_loop + the config fields it reads."""
    flat, off = flatten(sequences)
    return train_flat(flat, off, vocab_size, min_frequency, special_tokens)


def merge_loop_recount_py(sequences, vocab_size: int, min_frequency: int, special_tokens):
    """Pure-Python full-recount restatement (SURVEY Appendix A rules 1-9) for SMALL cases only.

    Independent of the C code above: recounts every pair from scratch each iteration, which is the
    form the HIP path's `recount` mode takes.  Used to cross-check the C oracle and the GPU path."""
    vocab: dict[bytes, int] = {bytes([b]): b for b in range(256)}            # trainer.py:123-125
    for t in special_tokens:                                                  # trainer.py:128-132
        tb = t.encode("utf-8") if isinstance(t, str) else bytes(t)
        if tb not in vocab:
            vocab[tb] = len(vocab)
    words = Counter(tuple(bytes([b]) for b in s) for s in sequences)          # trainer.py:221-225
    merges: list[tuple[bytes, bytes]] = []
    for _ in range(max(0, vocab_size - len(vocab))):                          # trainer.py:238-241
        pc: Counter = Counter()
        for w, f in words.items():                                            # rule 3: every adjacent position
            for j in range(len(w) - 1):
                pc[(w[j], w[j + 1])] += f
        if not pc:
            break
        best = max(pc.items(), key=lambda kv: (kv[1], kv[0]))[0]              # trainer.py:246
        if pc[best] < min_frequency:
            break
        x, y = best
        z = x + y
        new_words: Counter = Counter()
        for w, f in words.items():                                            # rule 7: greedy left to right
            out, j = [], 0
            while j < len(w):
                if j + 1 < len(w) and w[j] == x and w[j + 1] == y:
                    out.append(z)
                    j += 2
                else:
                    out.append(w[j])
                    j += 1
            new_words[tuple(out)] += f
        words = new_words
        merges.append(best)
        if z not in vocab:                                                    # trainer.py:298-300
            vocab[z] = len(vocab)
    return vocab, merges


def merges_hex(merges) -> str:
    """Serialisation used by the golden files and BASELINE.md's sha256 pins."""
    return "".join(f"{a.hex()} {b.hex()}\n" for a, b in merges)
