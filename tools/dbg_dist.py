#!/usr/bin/env python3
"""Dev tool: one multi-rank scenario of tests/test_gpu_distributed.py with option overrides; prints where the merges first differ from the oracle.
   python tools/dbg_dist.py synthetic_small_buffers 2 batch_max=1 verify=0"""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
from tests import dist_workers
from tests.test_gpu_distributed import _expect


def worker(rank, world, dist, scenario, extra):
    import tests.dist_workers as dw
    from yet_another_bpe import distributed
    orig = distributed.train_sharded
    def patched(factory, flat, off, freq, base, merges, minf, rank, world, transport="rccl", options=None):
        o = dict(options or {}); o.update(extra)
        return orig(factory, flat, off, freq, base, merges, minf, rank, world, transport=transport, options=o)
    distributed.train_sharded = patched
    try:
        return dw.gpu_sharded(rank, world, dist, scenario)
    finally:
        distributed.train_sharded = orig


if __name__ == "__main__":
    scenario, world = sys.argv[1], int(sys.argv[2])
    extra = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[3:])}
    exp = _expect(scenario)
    try:
        outs = dist_workers.spawn(worker, world, scenario, extra, timeout=600)
    except AssertionError as e:
        print("FAILED:", str(e)[-600:]); sys.exit(1)
    for r, (merges, n_words, rebuilds, retiles) in enumerate(outs):
        got = [(bytes.fromhex(a), bytes.fromhex(b)) for a, b in merges]
        first = next((i for i, (g, x) in enumerate(zip(got, exp)) if g != x), None)
        print(f"rank {r}: {len(got)} merges (expected {len(exp)}), first mismatch at {first}, rebuilds {rebuilds} retiles {retiles}")
