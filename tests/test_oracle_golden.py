"""CPU: the oracle (oracle/bpe_oracle.c + the pure-Python recount) against the golden vectors.

G2 is the reference's own fixture; G1/G3/G4/G5 were produced by running the reference (tests/golden/make_golden.py).
"""
from __future__ import annotations

import hashlib
import json
import pickle
import random

import numpy as np
import pytest

from oracle import oracle
from tests import helpers

SP = ["<|endoftext|>"]


def test_g2_reference_fixture_243(golden_dir):
    """tests/test_train_bpe_gpt2.py:27-50 of the reference: merges on corpus.en at vocab_size=500."""
    expected = helpers.read_gpt2_merges(golden_dir / "g2_reference_merges_243.txt")
    assert len(expected) == 243
    vocab, merges = oracle.merge_loop(helpers.corpus_en_words(), 500, 1, SP)
    assert merges == expected
    assert len(vocab) == 500


def test_g1_corpus_en_exhaustive_and_sha_pins(golden_dir):
    g1 = helpers.read_hex_merges(golden_dir / "g1_corpus_en_exhaustive.hex")
    meta = json.loads((golden_dir / "g1_meta.json").read_text())
    assert len(g1) == meta["n_exhaustive"] == 8199
    vocab, merges = oracle.merge_loop(helpers.corpus_en_words(), 10 ** 6, 1, SP)
    assert merges == g1
    for k, sha in meta["sha256"].items():  # the pins of BASELINE.md section 4
        assert hashlib.sha256(oracle.merges_hex(merges[: int(k)]).encode()).hexdigest() == sha
    # min_frequency=2 stops after 4,439 merges
    _, m2 = oracle.merge_loop(helpers.corpus_en_words(), 10 ** 6, 2, SP)
    assert m2 == g1[: meta["n_min_frequency_2"]]


def test_g1_vocab_ids_at_1000(golden_dir):
    ref_vocab = {bytes.fromhex(k): v for k, v in json.loads((golden_dir / "g1_corpus_en_vocab_1000.json").read_text()).items()}
    vocab, merges = oracle.merge_loop(helpers.corpus_en_words(), 1000, 1, SP)
    assert vocab == ref_vocab
    assert len(merges) == 743


@pytest.mark.parametrize("impl", ["c", "py"])
def test_g345_cases(impl):
    for c in helpers.golden_cases():
        if impl == "py" and sum(len(w) for w in c["words_b"]) > 400:
            continue
        fn = oracle.merge_loop if impl == "c" else oracle.merge_loop_recount_py
        vocab, merges = fn(c["words_b"], c["vocab_size"], c["min_frequency"], c["special_tokens"])
        assert merges == c["merges_b"], c["name"]
        assert len(vocab) == c["vocab_len"], c["name"]
        assert {k: v for k, v in vocab.items() if v >= 256} == c["vocab_b"], c["name"]


def test_g5_config2_synthetic(golden_dir):
    """BASELINE configs[1]: 10 MiB synthetic ASCII, 1k merges."""
    from yet_another_bpe import synth

    meta = json.loads((golden_dir / "g5_meta.json").read_text())
    flat, off = synth.generate(synth.SynthSpec.config2())
    assert hashlib.sha256(flat.tobytes()).hexdigest() == meta["corpus_sha256"]
    assert hashlib.sha256(off.tobytes()).hexdigest() == meta["offsets_sha256"]
    vocab, merges = oracle.train_flat(flat, off, 257 + 1000, 1, SP)
    assert merges == helpers.read_hex_merges(golden_dir / "g5_config2_merges_1000.hex")
    assert hashlib.sha256(oracle.merges_hex(merges).encode()).hexdigest() == meta["merges_sha256"]


def test_g7_snapshot_is_consistent(golden_dir):
    """The reference's snapshot (tests/_snapshots/test_train_bpe_special_tokens.pkl) cannot be re-run (its corpus
    blob is missing from the reference checkout), but its internal structure pins rule A-6: a merge that
    re-creates the special token's bytes consumes no id.  Primitive-opcode pickle."""
    with open(golden_dir / "snapshot_special_tokens.pkl", "rb") as f:
        snap = pickle.load(f)
    assert len(snap["merges"]) == 743 and len(snap["vocab_values"]) == 999
    assert (b"<", b"|endoftext|>") in snap["merges"]
    toks = {bytes([b]) for b in range(256)} | {b"<|endoftext|>"}
    for a, b in snap["merges"]:
        assert a in toks and b in toks
        toks.add(a + b)
    assert toks == snap["vocab_values"]


def test_c_oracle_equals_python_recount_random():
    rng = random.Random(7)
    for t in range(150):
        al = rng.choice([b"ab", b"abc", b"xyz ", bytes([0, 255, 254, 1])])
        words = []
        for _ in range(rng.randint(0, 7)):
            words += [bytes(rng.choice(al) for _ in range(rng.randint(1, 10)))] * rng.randint(1, 4)
        sp = rng.choice([[], ["ab"], ["<|x|>"], ["a", "b"]])
        vs, mf = 256 + rng.randint(0, 30), rng.randint(1, 3)
        assert oracle.merge_loop(words, vs, mf, sp) == oracle.merge_loop_recount_py(words, vs, mf, sp)
