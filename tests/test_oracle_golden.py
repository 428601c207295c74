"""CPU: the oracle (oracle/bpe_oracle.c + the pure-Python recount) against the golden vectors.

G2 is the reference's own fixture; G1/G3/G4/G5 were produced by running the reference (tests/golden/make_golden.py).
"""
from __future__ import annotations

import hashlib
import io
import json
import pickle
import random

import numpy as np
import pytest

from oracle import oracle, py_trainer
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
    class NoGlobals(pickle.Unpickler):  # the file is a copy of untrusted content: primitives only, no callables
        def find_class(self, module, name):
            raise pickle.UnpicklingError(f"refusing global {module}.{name}")

    snap = NoGlobals(io.BytesIO((golden_dir / "snapshot_special_tokens.pkl").read_bytes())).load()
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


# ---- the pure-Python restatement with the reference's data structures (oracle/py_trainer.py: bench.py's CPU baseline)
def test_py_trainer_g1_prefix_and_vocab_ids(golden_dir):
    g1 = helpers.read_hex_merges(golden_dir / "g1_corpus_en_exhaustive.hex")
    ref_vocab = {bytes.fromhex(k): v for k, v in json.loads((golden_dir / "g1_corpus_en_vocab_1000.json").read_text()).items()}
    vocab, merges = py_trainer.merge_loop(helpers.corpus_en_words(), 1000, 1, SP)
    assert merges == g1[:743] and vocab == ref_vocab
    _, m2 = py_trainer.merge_loop(helpers.corpus_en_words(), 257 + 2000, 2, SP)
    assert m2 == g1[:2000]  # (min_frequency=2 only bites after 4,439 merges)


def test_py_trainer_g345_cases():
    for c in helpers.golden_cases():
        vocab, merges = py_trainer.merge_loop(c["words_b"], c["vocab_size"], c["min_frequency"], c["special_tokens"])
        assert merges == c["merges_b"], c["name"]
        assert len(vocab) == c["vocab_len"], c["name"]
        assert {k: v for k, v in vocab.items() if v >= 256} == c["vocab_b"], c["name"]


def test_py_trainer_g5_config2_prefix(golden_dir):
    """BASELINE configs[1] (10 MiB synthetic ASCII): the first 120 merges (the whole 1,000 take the reference ~7 s)."""
    from yet_another_bpe import synth

    flat, off = synth.generate(synth.SynthSpec.config2())
    fb, o = flat.tobytes(), off.tolist()
    words = [fb[o[i]:o[i + 1]] for i in range(len(o) - 1)]
    _, merges = py_trainer.merge_loop(words, 257 + 120, 1, SP)
    assert merges == helpers.read_hex_merges(golden_dir / "g5_config2_merges_1000.hex")[:120]


def test_py_trainer_time_cap_returns_a_prefix():
    words = helpers.corpus_en_words()
    _, full = py_trainer.merge_loop(words, 257 + 400, 1, SP)
    _, part = py_trainer.merge_loop(words, 257 + 400, 1, SP, max_seconds=0.05)
    assert part == full[: len(part)] and len(part) < len(full)
