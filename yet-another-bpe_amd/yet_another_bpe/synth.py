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


# ---------------------------------------------------------------- synthetic TEXT (BASELINE configs[4] shape)
# A lexicon of text pieces -- the same integer-only construction on the build host (where the expected values are made) and on the GPU box -- drawn
# with the Zipf weights above and concatenated (device: yabpe_synth_generate_lex; host: generate_lex below).  What a piece
# is decides which parts of the GPT-2 pattern (reference trainer.py:163) the text exercises:
#   latin words with a leading space, Cyrillic (2-byte UTF-8) and CJK (3-byte) words, Gothic letters and emoji (4-byte),
#   digit runs, punctuation runs, contractions, whitespace runs with U+00A0 / U+3000 / tabs / newlines, long letter and
#   space runs (64..300 bytes: pre-tokens that do not fit the tile stream: the long-word path) and the special token itself.
_LATIN = b"etaoinshrdlucmfwypvbgkjqxzETAOINSHR"
_PUNCT = b".,;:!?-()\"/*&%#@"
_CONTR = (b"'s", b"'t", b"'re", b"'ve", b"'ll", b"'d", b"'m", b"'S", b"'x")
_WS = (" ", "\n", "\t", "\r\n", "\u00a0", "\u3000", "  ", "\n\n")


def text_lexicon(n_types: int, seed: int, special: bytes = b"<|endoftext|>") -> tuple[np.ndarray, np.ndarray]:
    """-> (lexicon bytes u8, offsets u64[n_types + 1]).  Entry j depends on (seed, j) only."""
    j = np.arange(n_types, dtype=np.uint64)
    kind = (rnd(seed, 4, j) % np.uint64(1000)).astype(np.int64)
    r5 = rnd(seed, 5, j)
    out: list[bytes] = []
    for t in range(n_types):
        k, r = int(kind[t]), int(r5[t])
        ch = rnd(seed, 6, np.uint64(t) * np.uint64(512) + np.arange(300, dtype=np.uint64))  # this entry's character stream

        def pick(alpha, n, ch=ch):
            return [alpha[int(c % np.uint64(len(alpha)))] for c in ch[:n]]

        if k < 560:    # " word"
            e = b" " + bytes(pick(_LATIN, 1 + r % 10))
        elif k < 650:  # Cyrillic word, leading space
            e = (" " + "".join(chr(0x0430 + int(c % np.uint64(32))) for c in ch[:1 + r % 8])).encode("utf-8")
        elif k < 700:  # CJK run, no space
            e = "".join(chr(0x4E00 + int(c % np.uint64(2000))) for c in ch[:1 + r % 4]).encode("utf-8")
        elif k < 715:  # Gothic letters (4-byte \p{L})
            e = (" " + "".join(chr(0x10330 + int(c % np.uint64(20))) for c in ch[:1 + r % 3])).encode("utf-8")
        elif k < 730:  # emoji (4-byte, neither letter nor number nor space)
            e = "".join(chr(0x1F600 + int(c % np.uint64(40))) for c in ch[:1 + r % 2]).encode("utf-8")
        elif k < 810:  # number, leading space
            e = b" " + bytes(0x30 + int(c % np.uint64(10)) for c in ch[:1 + r % 6])
        elif k < 880:  # punctuation run
            e = bytes(pick(_PUNCT, 1 + r % 3))
        elif k < 920:  # whitespace run
            e = "".join(_WS[int(c % np.uint64(len(_WS)))] for c in ch[:1 + r % 4]).encode("utf-8")
        elif k < 950:  # contraction (attaches to the piece before it)
            e = _CONTR[r % len(_CONTR)]
        elif k < 975:  # the special token, as text
            e = special
        elif t >= 2000 and k < 985:  # long letter run (one pre-token of 64..300 bytes)
            e = b" " + bytes(pick(_LATIN[:12], 64 + r % 237))
        elif t >= 2000 and k < 990:  # long run of spaces
            e = b" " * (64 + r % 237)
        elif t >= 2000 and k < 995:  # long digit run
            e = bytes(0x30 + int(c % np.uint64(10)) for c in ch[:64 + r % 237])
        else:
            e = b" " + bytes(pick(_LATIN, 2 + r % 6))
        out.append(e)
    off = np.zeros(n_types + 1, dtype=np.uint64)
    np.cumsum([len(e) for e in out], out=off[1:])
    return np.frombuffer(b"".join(out), dtype=np.uint8).copy(), off


def generate_lex(target_bytes: int, seed: int, lex_bytes: np.ndarray, lex_off: np.ndarray, chunk: int = 1 << 22) -> np.ndarray:
    """Host twin of yabpe_synth_generate_lex: the text (u8).  Draw i -> type = first j with cum[j] > rnd(seed,3,i) % sum(w)."""
    n_types = len(lex_off) - 1
    cum = zipf_cum(n_types)
    total = cum[-1]
    ll = np.diff(lex_off).astype(np.int64)
    pieces, nbytes, i0 = [], 0, 0
    while nbytes < target_bytes:
        i = np.arange(i0, i0 + chunk, dtype=np.uint64)
        ty = np.searchsorted(cum, rnd(seed, 3, i) % total, side="right").astype(np.int64)
        wl = ll[ty]
        c = np.cumsum(wl)
        need = target_bytes - nbytes
        if int(c[-1]) >= need:
            k = int(np.searchsorted(c, need, side="left")) + 1
            ty, wl, c = ty[:k], wl[:k], c[:k]
        n_here = int(c[-1])
        starts = c - wl
        # byte p of this chunk belongs to draw d = searchsorted(c, p, right); its source is lex_off[ty[d]] + (p - starts[d])
        d = np.repeat(np.arange(len(ty), dtype=np.int64), wl)
        src = lex_off[ty].astype(np.int64)[d] + (np.arange(n_here, dtype=np.int64) - starts[d])
        pieces.append(lex_bytes[src])
        nbytes += n_here
        i0 += chunk
    return np.concatenate(pieces)
