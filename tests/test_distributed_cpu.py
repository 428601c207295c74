"""CPU (gloo, world_size 2): shard planning, the all-gather callback plumbing of the custom transport, and a pure-Python
model of the multi-GPU protocol (word shards + replicated pair table + per-merge delta exchange) against the oracle."""
from __future__ import annotations

import numpy as np
import pytest

from oracle import oracle
from tests import dist_workers, helpers
from yet_another_bpe.distributed import plan_shards, plan_shards_device


def test_plan_shards_covers_and_balances():
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 40, size=5000).astype(np.uint64)
    off = np.zeros(len(lens) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    for world in (1, 2, 3, 4, 8):
        sh = plan_shards(off, world)
        assert sh[0][0] == 0 and sh[-1][1] == len(lens)
        assert all(sh[i][1] == sh[i + 1][0] for i in range(world - 1))
        load = [int(off[b]) - int(off[a]) + (b - a) for a, b in sh]
        assert max(load) - min(load) <= 2 * 41  # balanced to within a word or two
        assert plan_shards_device(lambda i: int(off[i]), len(lens), world) == sh
    # a slice of a larger corpus (absolute offsets) plans the same way
    sub = off[100:2101]
    assert plan_shards(sub, 2) == plan_shards(sub - sub[0], 2)
    # degenerate: fewer words than ranks
    tiny = np.array([0, 3], dtype=np.uint64)
    sh = plan_shards(tiny, 4)
    assert sum(b - a for a, b in sh) == 1


def test_transport_callback_gloo_world2():
    sums = dist_workers.spawn(dist_workers.cpu_transport_roundtrip, 2)
    assert sums[0] == sums[1]


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_sharded_protocol_equals_single_rank(world):
    words = [b"low"] * 5 + [b"lower"] * 2 + [b"widest"] * 3 + [b"newest"] * 6 + [b"abab"] * 4 + [b"aaaa", b"aaa", b"<|endoftext|>"] * 2
    sp = ["<|endoftext|>"]
    exp_vocab, exp_merges = oracle.merge_loop(words, 257 + 40, 1, sp)
    outs = dist_workers.spawn(dist_workers.cpu_sharded_reference, world, [w.hex() for w in words], sp, 257 + 40, 1)
    for merges, vocab_len in outs:
        assert [(bytes.fromhex(a), bytes.fromhex(b)) for a, b in merges] == exp_merges
        assert vocab_len == len(exp_vocab)


def test_plan_chunk_shards_partitions_and_balances():
    from yet_another_bpe.distributed import plan_chunk_shards

    rng = np.random.default_rng(4)
    for world in (1, 2, 3, 8):
        for n in (0, 1, 2, 7, 50):
            sizes = rng.integers(1, 1000, size=n).tolist()
            plan = plan_chunk_shards(sizes, world)
            assert len(plan) == world and plan[0][0] == 0 and plan[-1][1] == n
            assert all(a <= b for a, b in plan) and all(plan[i][1] == plan[i + 1][0] for i in range(world - 1))
            if n >= 4 * world:
                loads = [sum(sizes[a:b]) for a, b in plan]
                assert max(loads) - min(loads) <= 2 * max(sizes)
