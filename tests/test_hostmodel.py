"""CPU: the per-position merge rules the HIP kernels use (yet-another-bpe_amd/csrc/tile_logic.h), driven by a
sequential model of the device data layout (tests/hostmodel/tile_model.cpp), against the oracle / golden vectors.
The model re-counts the whole stream after every merge and requires the incrementally updated table to match."""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

from tests import helpers

HM = Path(__file__).resolve().parent / "hostmodel"


@pytest.fixture(scope="module")
def model_lib():
    so = HM / "libtile_model.so"
    src = HM / "tile_model.cpp"
    hdr = HM.parent.parent / "yet-another-bpe_amd/csrc/tile_logic.h"
    if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, hdr.stat().st_mtime):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(so), str(src)])
    lib = ctypes.CDLL(str(so))
    lib.tile_model_train.restype = ctypes.c_int
    return lib


def run_model(lib, words, freq, vocab_size, min_frequency, specials, cap, lmax, verify=1):
    toks = helpers.base_tokens(specials)
    flat, off = helpers.flatten(words)
    tb = np.frombuffer(b"".join(toks), dtype=np.uint8).copy()
    to = np.zeros(len(toks) + 1, dtype=np.uint32)
    to[1:] = np.cumsum([len(t) for t in toks])
    nm = max(0, vocab_size - len(toks))
    L, R, M = (np.zeros(nm + 1, np.uint32) for _ in range(3))
    C = np.zeros(nm + 1, np.uint64)
    fq = None if freq is None else np.ascontiguousarray(freq, dtype=np.uint64)
    vp = ctypes.c_void_p
    n = lib.tile_model_train(vp(flat.ctypes.data if flat.size else 0), vp(off.ctypes.data), vp(fq.ctypes.data if fq is not None else 0),
                             ctypes.c_uint64(len(words)), vp(tb.ctypes.data), vp(to.ctypes.data), ctypes.c_uint32(len(toks)),
                             ctypes.c_uint32(nm), ctypes.c_uint64(min_frequency), cap, lmax, verify,
                             vp(L.ctypes.data), vp(R.ctypes.data), vp(M.ctypes.data), vp(C.ctypes.data))
    assert n >= 0, f"model failed / table mismatch at iteration {-1 - n}"
    t = list(toks)
    merges = []
    for i in range(n):
        merges.append((t[L[i]], t[R[i]]))
        if M[i] == len(t):
            t.append(t[L[i]] + t[R[i]])
        else:
            assert t[M[i]] == t[L[i]] + t[R[i]]
    return {b: i for i, b in enumerate(t)}, merges


@pytest.mark.parametrize("cap,lmax", [(64, 16), (1024, 64), (32, 8)])
def test_rules_on_golden_cases(model_lib, cap, lmax):
    for c in helpers.golden_cases():
        v, m = run_model(model_lib, c["words_b"], None, c["vocab_size"], c["min_frequency"], c["special_tokens"], cap, lmax)
        assert m == c["merges_b"] and len(v) == c["vocab_len"], ("flat", c["name"])
        uw, fq = helpers.pooled(c["words_b"])
        v, m = run_model(model_lib, uw, fq, c["vocab_size"], c["min_frequency"], c["special_tokens"], cap, lmax)
        assert m == c["merges_b"] and len(v) == c["vocab_len"], ("weighted", c["name"])


def test_rules_on_corpus_en(model_lib, golden_dir):
    g1 = helpers.read_hex_merges(golden_dir / "g1_corpus_en_exhaustive.hex")
    words = helpers.corpus_en_words()
    _, m = run_model(model_lib, words, None, 257 + 1500, 1, ["<|endoftext|>"], 1024, 64, verify=0)
    assert m == g1[:1500]
    uw, fq = helpers.pooled(words)
    _, m = run_model(model_lib, uw, fq, 10 ** 6 if False else 257 + 8300, 1, ["<|endoftext|>"], 1024, 64, verify=0)
    assert m == g1
