"""GPU, sizes and shapes BETWEEN the pinned fixtures: synthetic corpora of several sizes, alphabets and seeds, flat and pooled
layouts, with the forms of the sparse launch forced (8- and 16-wave workgroups, small batches, a tight selection window)
-- every run against the C oracle on the same words (bit-exact id triples) and against a recount of its own final stream
(`verify_table`).  The oracle is the checker here, as everywhere in tests/."""
from __future__ import annotations

import numpy as np
import pytest

from oracle import oracle
from tests import helpers

pytestmark = pytest.mark.gpu
SP = ["<|endoftext|>"]

CASES = [
    # (MiB, types, seed, alphabet, space prefix, merges, dedup, options)
    (24, 20_000, 11, b"abcdefghijklmnopqrstuvwxyz", True, 6000, False, {}),
    (24, 20_000, 11, b"abcdefghijklmnopqrstuvwxyz", True, 6000, True, {}),
    (48, 200_000, 12, bytes(range(256)), False, 8000, False, {"full_wpb": 16}),
    (48, 200_000, 12, bytes(range(256)), False, 8000, False, {"full_wpb": 8, "batch_max": 3}),
    (32, 3_000, 13, b"abcdefgh", True, 4000, False, {"check_interval": 7}),          # few word types: long chains, many ties
    (32, 3_000, 13, b"abcdefgh", True, 4000, True, {"batch_max": 16, "cand_target": 128}),
    (64, 1_000_000, 14, bytes(range(256)), False, 12000, False, {"retile_pct": 90}),  # retiles often
]


IDS = ["ascii_24mib_flat", "ascii_24mib_pooled", "bytes_48mib_16waves", "bytes_48mib_8waves_batch3", "few_types_flat_check7", "few_types_pooled_short_list", "bytes_64mib_retiles"]


@pytest.mark.parametrize("mib,types,seed,alphabet,prefix,merges,dedup,opts", CASES, ids=IDS)
def test_between_the_fixtures(mib, types, seed, alphabet, prefix, merges, dedup, opts):
    from yet_another_bpe import _native, synth

    spec = synth.SynthSpec(mib << 20, types, seed, alphabet, prefix)
    flat, off = synth.generate(spec)
    _vocab, _merges, ids = oracle.train_flat(flat, off, 257 + merges, 1, SP, return_ids=True)
    with _native.Context() as ctx:
        for k, v in opts.items():
            ctx.set_option(k, v)
        ctx.set_vocab(helpers.base_tokens(SP))
        ctx.load_words(flat, off, None, dedup=dedup)
        left, right, merged, count = ctx.train(merges, 1)
        assert ctx.verify_table() == 0
    n = len(ids["left"])
    assert len(left) == n
    assert np.array_equal(left, ids["left"]) and np.array_equal(right, ids["right"]) and np.array_equal(merged, ids["merged"])
    assert np.array_equal(np.asarray(count, dtype=np.uint64), np.asarray(ids["count"], dtype=np.uint64))
