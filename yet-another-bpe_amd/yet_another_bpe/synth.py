"""Synthetic word corpus of SURVEY.md §8(d): integer-only, so host (here), and device
(csrc/yabpe_synth.hip, `yabpe_synth_*`) produce identical bytes.

    mix(x)            splitmix64 finalizer
    rnd(seed, s, i) = mix(seed + 0x9E3779B97F4A7C15*(i+1) + 0xD1B54A32D192ED03*s)      (u64 wraparound)
    lexicon           len_j = 1 + rnd(seed,1,j) % 12 ; byte k of type j = alphabet[rnd(seed,2,16*j+k) % |alphabet|]
    Zipf(s=1)         w_j = floor(2**40/(j+1)) ; u = rnd(seed,3,i) % sum(w) ; type = first j with cum[j] > u
    corpus            word i = type drawn for i (optionally prefixed with one 0x20 byte);
                      words are emitted until the byte total reaches `target_bytes`; the last word is not truncated.

The trainer receives the words directly (flat bytes + offsets) -- no regex, no UTF-8 constraint.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)
_S = np.uint64(0xD1B54A32D192ED03)


def mix(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    x ^= x >> np.uint64(30)
    x *= _M1
    x ^= x >> np.uint64(27)
    x *= _M2
    x ^= x >> np.uint64(31)
    return x


def rnd(seed: int, stream: int, i: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        i = np.asarray(i, dtype=np.uint64)
        return mix(np.uint64(seed) + _G * (i + np.uint64(1)) + _S * np.uint64(stream))


@dataclass(frozen=True)
class SynthSpec:
    target_bytes: int
    n_types: int
    seed: int
    alphabet: bytes
    space_prefix: bool

    @staticmethod
    def config2() -> "SynthSpec":
        """BASELINE configs[1]: 10 MiB ASCII, 50k types, every word prefixed with 0x20."""
        return SynthSpec(10 << 20, 50_000, 2, b"abcdefghijklmnopqrstuvwxyz", True)

    @staticmethod
    def config3(target_bytes: int = 1 << 30) -> "SynthSpec":
        """BASELINE configs[2]: 1 GiB, all 256 byte values, 1M types."""
        return SynthSpec(target_bytes, 1_000_000, 3, bytes(range(256)), False)


def lexicon(spec: SynthSpec) -> tuple[np.ndarray, np.ndarray]:
    """-> (type_bytes u8[V,12], type_len u8[V])  (length without the optional space prefix)."""
    j = np.arange(spec.n_types, dtype=np.uint64)
    tlen = (1 + rnd(spec.seed, 1, j) % np.uint64(12)).astype(np.uint8)
    idx = (j[:, None] * np.uint64(16) + np.arange(12, dtype=np.uint64)[None, :])
    alpha = np.frombuffer(spec.alphabet, dtype=np.uint8)
    tb = alpha[(rnd(spec.seed, 2, idx) % np.uint64(len(alpha))).astype(np.int64)]
    return tb.astype(np.uint8), tlen


def zipf_cum(n_types: int) -> np.ndarray:
    w = (np.uint64(1) << np.uint64(40)) // (np.arange(n_types, dtype=np.uint64) + np.uint64(1))
    return np.cumsum(w, dtype=np.uint64)


def generate(spec: SynthSpec, chunk_words: int = 1 << 22) -> tuple[np.ndarray, np.ndarray]:
    """-> (flat u8 bytes, u64 offsets[n_words+1])."""
    tb, tlen = lexicon(spec)
    cum = zipf_cum(spec.n_types)
    total = cum[-1]
    pre = 1 if spec.space_prefix else 0
    wl_all, ty_all = [], []
    nbytes, i0 = 0, 0
    while nbytes < spec.target_bytes:
        i = np.arange(i0, i0 + chunk_words, dtype=np.uint64)
        u = rnd(spec.seed, 3, i) % total
        ty = np.searchsorted(cum, u, side="right").astype(np.int64)
        wl = tlen[ty].astype(np.uint64) + np.uint64(pre)
        c = np.cumsum(wl, dtype=np.uint64)
        need = spec.target_bytes - nbytes
        if int(c[-1]) >= need:
            k = int(np.searchsorted(c, np.uint64(need), side="left")) + 1  # first k words reaching the target
            ty, wl = ty[:k], wl[:k]
            nbytes += int(c[k - 1])
        else:
            nbytes += int(c[-1])
        wl_all.append(wl)
        ty_all.append(ty)
        i0 += chunk_words
    wl = np.concatenate(wl_all)
    ty = np.concatenate(ty_all)
    off = np.zeros(len(wl) + 1, dtype=np.uint64)
    np.cumsum(wl, out=off[1:])
    flat = np.empty(int(off[-1]), dtype=np.uint8)
    # scatter type bytes: position k of every word
    starts = off[:-1].astype(np.int64)
    if pre:
        flat[starts] = 0x20
    tl = tlen[ty].astype(np.int64)
    for k in range(12):
        sel = tl > k
        flat[starts[sel] + pre + k] = tb[ty[sel], k]
    return flat, off
