#!/usr/bin/env python3
"""Dev tool: one config-3 job (1 GiB, 32k merges by default) with stats and a per-segment time table.
   python tools/quick_job.py [--merges N] [--mib M] [--sample S] [--opt k=v ...] [--dedup]"""
import argparse, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
import numpy as np
from yet_another_bpe import _native, synth
ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=1024)
ap.add_argument("--merges", type=int, default=32000)
ap.add_argument("--sample", type=int, default=8)
ap.add_argument("--runs", type=int, default=2)
ap.add_argument("--dedup", action="store_true")
ap.add_argument("--opt", action="append", default=[])
ap.add_argument("--dump-iter", default="", help="write per-merge algorithmic / actual stream bytes of the last run (JSON)")
a = ap.parse_args()
spec = synth.SynthSpec.config3(a.mib << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    for run in range(a.runs):
        with _native.Context() as ctx:
            ctx.set_option("event_sample", a.sample if run == a.runs - 1 else 0)
            for kv in a.opt:
                k, v = kv.split("=")
                ctx.set_option(k, int(v))
            ctx.set_vocab(base)
            ctx.load_words_ptr(pb, po, nw, dedup=a.dedup)
            t0 = time.perf_counter()
            left, right, merged, count = ctx.train(a.merges, 1)
            wall = time.perf_counter() - t0
            st = ctx.stats()
            it, us, scan = ctx.event_log()
            sites, live = ctx.iter_log()
        print(f"run {run}: {len(left)} merges, wall {wall*1e3:.1f} ms, train_ms {st['train_ms']:.1f}, {len(left)/wall:.0f} merges/s | fused {st['fused_launches']} "
              f"cand_rebuilds {st['cand_rebuilds']} rescans {st['cand_rescans']} retiles {st['retiles']} table_rebuilds {st['table_rebuilds']} cap {st['table_capacity']} entries {st['table_entries']} | sparse: {st['sparse_merges']} merges in {st['sparse_launches']} launches "
              f"(mean batch {st['sparse_merges'] / max(1, st['sparse_launches']):.2f}), {st['sparse_ms']:.1f} ms = {1e3 * st['sparse_ms'] / max(1, st['sparse_merges']):.2f} us per merge; second half {1e3 * st['tail_ms'] / max(1, st['tail_merges']):.2f} us per merge", flush=True)
if len(it):
    edges = [0, 50, 150, 300, 1000, 3000, 8000, 12000, 20000, 32000, 50000]
    it = np.asarray(it); us = np.asarray(us)
    tot = 0.0
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (it >= lo) & (it < hi)
        if m.any():
            n_in = min(hi, len(left)) - lo
            est = us[m].mean() * n_in / 1e3
            tot += est
            print(f"  merges {lo:6d}-{hi:6d}: sampled {m.sum():5d}  apply+select avg {us[m].mean():8.1f} us  p50 {np.median(us[m]):8.1f}  max {us[m].max():8.1f}  => ~{est:7.1f} ms")
    print(f"  sum of segment estimates {tot:.1f} ms (events add ~4-5 us per sampled launch)")

if a.dump_iter:
    import json
    T, algo, actual = st["tokens_initial"], [], []
    for k in range(len(left)):
        algo.append(int(2 * (T + st["n_words_input"])))
        actual.append(int(2 * int(live[k]) + 4 * st["n_tiles"]))
        T -= int(sites[k])
    json.dump({"algo_bytes": algo, "actual_bytes": actual}, open(a.dump_iter, "w"))
