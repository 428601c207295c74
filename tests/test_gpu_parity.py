"""GPU (-m gpu): the HIP path through the C ABI against the oracle and the committed golden vectors.

Bar: bit-exact merges lists AND vocab id assignment (integer / byte work, no tolerance).
Layouts covered: flat (every occurrence resident), weighted (host-pooled words + counts), device-side pooling.
"""
from __future__ import annotations

import hashlib
import json
import random

import numpy as np
import pytest

from oracle import oracle
from tests import helpers

pytestmark = pytest.mark.gpu
SP = ["<|endoftext|>"]


def gpu_train(words, freq, vocab_size, min_frequency, specials, dedup=False, options=None, want_stats=False):
    from yet_another_bpe import _native

    base = helpers.base_tokens(specials)
    flat, off = helpers.flatten(words)
    nm = max(0, vocab_size - len(base))
    opts = {"verify": 1}
    opts.update(options or {})
    return _native.train_words(flat, off, freq, base, nm, min_frequency, dedup=dedup, options=opts, want_stats=want_stats)


def test_device_present_and_library_loaded():
    from yet_another_bpe import _native

    assert _native.lib().yabpe_device_count() >= 1


@pytest.mark.parametrize("layout", ["flat", "weighted", "device_dedup"])
def test_golden_cases(layout):
    for c in helpers.golden_cases():
        if not c["words_b"]:
            continue
        if layout == "weighted":
            uw, fq = helpers.pooled(c["words_b"])
            vocab, merges = gpu_train(uw, fq, c["vocab_size"], c["min_frequency"], c["special_tokens"])
        else:
            vocab, merges = gpu_train(c["words_b"], None, c["vocab_size"], c["min_frequency"], c["special_tokens"],
                                      dedup=(layout == "device_dedup"))
        assert merges == c["merges_b"], (layout, c["name"])
        assert len(vocab) == c["vocab_len"], (layout, c["name"])
        assert {k: v for k, v in vocab.items() if v >= 256} == c["vocab_b"], (layout, c["name"])


@pytest.mark.parametrize("layout", ["flat", "weighted", "device_dedup"])
def test_corpus_en_exhaustive_g1(golden_dir, layout):
    """BASELINE configs[0] and beyond: corpus.en until no pair is left (8,199 merges), min_frequency=1."""
    g1 = helpers.read_hex_merges(golden_dir / "g1_corpus_en_exhaustive.hex")
    words = helpers.corpus_en_words()
    if layout == "weighted":
        uw, fq = helpers.pooled(words)
        vocab, merges = gpu_train(uw, fq, 257 + 9000, 1, SP)
    else:
        vocab, merges = gpu_train(words, None, 257 + 9000, 1, SP, dedup=(layout == "device_dedup"))
    assert len(merges) == 8199
    assert merges == g1
    meta = json.loads((golden_dir / "g1_meta.json").read_text())
    for k, sha in meta["sha256"].items():
        assert hashlib.sha256(oracle.merges_hex(merges[: int(k)]).encode()).hexdigest() == sha


def test_corpus_en_config1_256_merges_and_vocab_ids(golden_dir):
    """BASELINE configs[0]: 256 bytes + 1 special + 256 merges => vocab_size 513; ids equal the reference's at 1000."""
    g1 = helpers.read_hex_merges(golden_dir / "g1_corpus_en_exhaustive.hex")
    words = helpers.corpus_en_words()
    _, merges = gpu_train(words, None, 513, 1, SP)
    assert merges == g1[:256]
    ref_vocab = {bytes.fromhex(k): v for k, v in json.loads((golden_dir / "g1_corpus_en_vocab_1000.json").read_text()).items()}
    vocab, merges = gpu_train(words, None, 1000, 1, SP)
    assert vocab == ref_vocab and merges == g1[:743]
    _, m2 = gpu_train(words, None, 257 + 9000, 2, SP)  # min_frequency=2 stops at 4,439
    assert m2 == g1[:4439]


def test_config2_synthetic_10mib_1k_merges(golden_dir):
    """BASELINE configs[1]: 10 MiB synthetic ASCII corpus, 1k merges, bit-exact vs the reference-made G5."""
    from yet_another_bpe import _native, synth

    spec = synth.SynthSpec.config2()
    meta = json.loads((golden_dir / "g5_meta.json").read_text())
    expected = helpers.read_hex_merges(golden_dir / "g5_config2_merges_1000.hex")
    base = helpers.base_tokens(SP)
    with _native.Context() as ctx:
        ctx.set_option("verify", 1)
        pb, po, nw, nb = ctx.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
        assert (nw, nb) == (meta["n_words"], meta["n_bytes"])
        # the device generator is bit-identical to the host one (and to what the reference was fed)
        assert hashlib.sha256(ctx.d2h(pb, nb).tobytes()).hexdigest() == meta["corpus_sha256"]
        assert hashlib.sha256(ctx.d2h(po, (nw + 1) * 8).tobytes()).hexdigest() == meta["offsets_sha256"]
        for dedup in (False, True):
            ctx.set_vocab(base)
            ctx.load_words_ptr(pb, po, nw, dedup=dedup)
            left, right, merged, count = ctx.train(1000, 1)
            toks = list(base)
            merges = []
            for l, r, m in zip(left.tolist(), right.tolist(), merged.tolist()):
                merges.append((toks[l], toks[r]))
                if m == len(toks):
                    toks.append(toks[l] + toks[r])
            assert merges == expected, f"dedup={dedup}"
            assert hashlib.sha256(oracle.merges_hex(merges).encode()).hexdigest() == meta["merges_sha256"]
            assert all(count[i] >= count[i + 1] for i in range(len(count) - 1))  # best count never increases


def test_random_small_vs_oracle():
    rng = random.Random(11)
    for t in range(120):
        al = rng.choice([b"ab", b"abc", b"xyz ", bytes([0, 255, 254, 1]), b"abcdefghijklmnop"])
        words = []
        for _ in range(rng.randint(1, 40)):
            words += [bytes(rng.choice(al) for _ in range(rng.randint(1, 30)))] * rng.randint(1, 6)
        rng.shuffle(words)
        sp = rng.choice([[], ["ab"], ["<|x|>"], ["a", "b"], ["[PAD]", "[UNK]", "[BOS]", "[EOS]"]])
        vs, mf = 256 + rng.randint(0, 120), rng.randint(1, 3)
        exp = oracle.merge_loop(words, vs, mf, sp)
        assert gpu_train(words, None, vs, mf, sp) == exp, t
        uw, fq = helpers.pooled(words)
        assert gpu_train(uw, fq, vs, mf, sp) == exp, t


def test_long_words_and_runs():
    """Words longer than the tile path's limit (63 tokens) take the long-word path; includes a==b runs."""
    rng = random.Random(5)
    words = [b" " * 1500, b"  ", b"x" + b" " * 301 + b"y", b"ab" * 5000, b"aaa", bytes(rng.choice(b"abc") for _ in range(9000))]
    words += [b"abcabc"] * 7 + [b"a" * 64, b"a" * 63, b"a" * 62, b"ab" * 31 + b"a", b"ab" * 32]
    exp = oracle.merge_loop(words, 257 + 150, 1, SP)
    assert gpu_train(words, None, 257 + 150, 1, SP) == exp
    uw, fq = helpers.pooled(words * 3)
    assert gpu_train(uw, fq, 257 + 150, 1, SP) == oracle.merge_loop(words * 3, 257 + 150, 1, SP)


def test_many_tiles_retile_and_table_growth():
    """A corpus large enough for several thousand tiles, with tiny table / frequent checks / forced retile,
    so that the halt-and-rebuild and retile service paths run; result must not depend on any of them."""
    from yet_another_bpe import synth

    spec = synth.SynthSpec(4 << 20, 20_000, 7, bytes(range(256)), False)
    flat, off = synth.generate(spec)
    base = helpers.base_tokens(SP)
    from yet_another_bpe import _native

    exp_vocab, exp_merges = oracle.train_flat(flat, off, 257 + 600, 1, SP)
    v1, m1, s1 = _native.train_words(flat, off, None, base, 600, 1, options={"verify": 1}, want_stats=True)
    assert (v1, m1) == (exp_vocab, exp_merges)
    v2, m2, s2 = _native.train_words(flat, off, None, base, 600, 1, want_stats=True,
                                     options={"verify": 1, "table_min_log2": 10, "check_interval": 7, "retile_pct": 95,
                                              "retile_min_tiles": 16, "apply_blocks": 3})
    assert (v2, m2) == (exp_vocab, exp_merges)
    assert s2["retiles"] >= 1
    # the always-sparse / never-sparse forms, every batch limit, unfused selection, ... give the same result
    for opts in ({"split": 1}, {"split": 0}, {"sig_rebuild_every": 3, "check_interval": 2}, {"rank_rides": 0}, {"cand_argmax": 0}, {"cand_min_count": 1, "check_interval": 3},
                 {"batch_max": 1}, {"batch_max": 2}, {"batch_max": 3, "check_interval": 5}, {"batch_max": 8, "check_interval": 1}, {"batch_max": 8, "split": 1, "check_interval": 3},
                 {"batch_max": 8, "cand_min_count": 1, "split": 1}, {"batch_max": 8, "cand_target": 64, "check_interval": 4}, {"batch_max": 8, "full_wpb": 16}, {"batch_max": 8, "full_wpb": 4, "agg_small": 0},
                 {"fuse_select": 0}, {"cand_rebuild_every": 1}, {"cand_rebuild_every": 100000, "check_interval": 8}, {"full_skip_blocks": 2}, {"full_skip_blocks": 1}, {"sig_rebuild_pct": 0, "check_interval": 5},
                 {"retile_pct": 95, "retile_min_tiles": 16}, {"split": 1, "retile_pct": 95, "retile_min_tiles": 16},
                 {"fused": 0}, {"fused": 0, "split": 1}, {"fused": 1, "check_interval": 3}, {"cand_target": 64, "check_interval": 4}, {"table_load_pct": 70, "table_grow_x": 2},
                 {"full_wpb": 4}, {"full_wpb": 8}, {"full_wpb": 16}, {"full_wpb": 16, "full_skip_blocks": 3}, {"full_wpb": 8, "check_interval": 5},
                 {"hist": 0}, {"hist": 0, "split": 0}, {"split": 0}, {"split": 0, "apply_blocks": 5, "check_interval": 3}):
        v3, m3 = _native.train_words(flat, off, None, base, 600, 1, options={"verify": 1, **opts})
        assert (v3, m3) == (exp_vocab, exp_merges), opts
    assert s1["scan_skip_launches"] > 0 and s1["scan_skip_tiles_read"] < s1["scan_skip_launches"] * s1["n_tiles"]
    assert s1["tokens_initial"] - s1["tokens_now"] == s2["tokens_initial"] - s2["tokens_now"]


def test_streaming_form_across_the_direct_store_limit():
    """The streaming launch keeps a workgroup's deltas in a direct-indexed LDS store while at most 512 tokens exist and in the
    hashed aggregator afterwards: a corpus over four letters stays dense (more sites than tiles) for hundreds of merges, so
    the switch happens inside the streaming phase; forced never-split as well.  Wide workgroups on the pooled layout too."""
    from yet_another_bpe import _native

    rng = np.random.default_rng(11)
    n_words = 120_000
    lens = rng.integers(2, 14, size=n_words)
    off = np.zeros(n_words + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    flat = rng.choice(np.frombuffer(b"abcd", dtype=np.uint8), size=int(off[-1]), p=[0.4, 0.3, 0.2, 0.1]).astype(np.uint8)
    base = helpers.base_tokens(SP)
    exp_vocab, exp_merges = oracle.train_flat(flat, off, 257 + 420, 1, SP)
    for opts in ({}, {"split": 0}, {"split": 0, "hist": 0}, {"split": 0, "check_interval": 7}, {"split": 1, "full_wpb": 16}):
        v, m, st = _native.train_words(flat, off, None, base, 420, 1, options={"verify": 1, **opts}, want_stats=True)
        assert (v, m) == (exp_vocab, exp_merges), opts
    for opts in ({"full_wpb": 16}, {"full_wpb": 8, "split": 1}):
        v, m = _native.train_words(flat, off, None, base, 420, 1, dedup=True, options={"verify": 1, **opts})
        assert (v, m) == (exp_vocab, exp_merges), opts


def test_weighted_layout_split_forms():
    """Pooled words + counts through the sparse form (batches of merges, wide and narrow workgroups, unfused) and the streaming one."""
    words = helpers.corpus_en_words()
    uw, fq = helpers.pooled(words)
    exp = oracle.merge_loop(words, 257 + 1200, 1, SP)
    for opts in ({"split": 1}, {"split": 1, "batch_max": 1}, {"split": 1, "batch_max": 2, "check_interval": 5}, {"split": 1, "full_wpb": 4},
                 {"split": 1, "fused": 0}, {"split": 0}, {"cand_min_count": 1, "split": 1, "check_interval": 3}):
        assert gpu_train(uw, fq, 257 + 1200, 1, SP, options={"verify": 1, **opts}) == exp, opts


def test_continue_training_equals_one_shot():
    from yet_another_bpe import _native

    words = helpers.corpus_en_words()
    base = helpers.base_tokens(SP)
    flat, off = helpers.flatten(words)
    with _native.Context() as ctx:
        ctx.set_vocab(base)
        ctx.load_words(flat, off)
        a = ctx.train(100, 1)
        b = ctx.train(150, 1)
        assert ctx.verify_table() == 0
    with _native.Context() as ctx:
        ctx.set_vocab(base)
        ctx.load_words(flat, off)
        c = ctx.train(250, 1)
    for i in range(3):
        assert np.array_equal(np.concatenate([a[i], b[i]]), c[i])


def test_vocab_id_space_limit_is_reported():
    """More merges than u16 ids: fine while the corpus runs out of pairs first (the reference returns normally there,
    e.g. vocab_size=100000 on a small corpus); a capacity error only when the loop really reaches the last id."""
    from yet_another_bpe import _native, synth

    with _native.Context() as ctx:
        ctx.set_vocab(helpers.base_tokens(SP))
        ctx.load_words(*helpers.flatten([b"ab", b"cd"]))
        left, right, merged, count = ctx.train(70_000, 1)
        assert len(left) == 2  # exhausted, no error
    flat, off = synth.generate(synth.SynthSpec(6 << 20, 120_000, 21, bytes(range(256)), False))  # > 65,277 merges possible
    with _native.Context() as ctx:
        ctx.set_vocab(helpers.base_tokens(SP))
        ctx.load_words(flat, off, dedup=True)
        with pytest.raises(_native.YabpeError) as e:
            ctx.train(70_000, 1)
        assert e.value.code == -4 and "id space exhausted" in str(e.value)


def test_many_long_words_filtered_by_signature():
    """100,000 words longer than the tile path's limit next to ordinary ones: the long-word launch tests one signature
    word per long word and rewrites only the words that may hold the pair (SURVEY 8f / VERDICT r1 #5); with and without
    the filter, and pooled, the result is the oracle's.  The time per merge of both forms goes to stdout (-s)."""
    import time

    from yet_another_bpe import _native

    rng = np.random.default_rng(17)
    n_long, n_short = 100_000, 60_000
    lens = np.concatenate([rng.integers(64, 91, size=n_long), rng.integers(2, 12, size=n_short)]).astype(np.uint64)
    rng.shuffle(lens)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    flat = rng.choice(np.frombuffer(b"abcdefghij  ", dtype=np.uint8), size=int(off[-1])).astype(np.uint8)
    base = helpers.base_tokens(SP)
    exp_vocab, exp_merges = oracle.train_flat(flat, off, 257 + 300, 1, SP)
    took = {}
    for opts in ({"long_sig": 1}, {"long_sig": 0}, {"long_sig": 1, "fused": 0}):
        t0 = time.perf_counter()
        v, m, st = _native.train_words(flat, off, None, base, 300, 1, options={"verify": 1, **opts}, want_stats=True)
        took[tuple(sorted(opts.items()))] = (time.perf_counter() - t0, st["train_ms"])
        assert (v, m) == (exp_vocab, exp_merges), opts
        assert st["n_long_words"] == n_long
    print("long words: train_ms per merge", {k: round(v[1] * 1000 / 300, 1) for k, v in took.items()}, "us")
    v, m = _native.train_words(flat, off, None, base, 300, 1, dedup=True, options={"verify": 1})
    assert (v, m) == (exp_vocab, exp_merges)
