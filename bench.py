#!/usr/bin/env python3
"""bench.py -- BPE training hot path on MI355X: merges/s (+ corpus bytes/s) to 32k merges on the 1 GiB
synthetic byte corpus of SURVEY.md 8(d) config 3, with the bound of each phase of the timed job -- the HBM roofline of
the streaming phase (fused k_apply, whole iterations) and the dependent-trip floor of the sparse phase (k_scan_skip: one
launch applies a BATCH of merges) -- and the reference's algorithm in pure Python on the host as the CPU baseline.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE whole training job: words already resident in HBM (flat u8 bytes + u64 offsets) ->
tile build + initial pair count -> `merges` merge iterations.  value = merges completed / wall time of the
K timed steps (max over ranks).  N > 1: the same 1 GiB corpus is word-sharded over the ranks (strong scaling).
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
for p in (REPO, REPO / "yet-another-bpe_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def _leading_words(ctx, pb, po, n_words, budget_bytes):
    import numpy as np

    off_all = ctx.d2h(po, min(n_words + 1, budget_bytes // 2 + 2) * 8, dtype=np.uint64)
    k = int(np.searchsorted(off_all, budget_bytes, side="right")) - 1
    k = max(1, min(k, len(off_all) - 1))
    off = off_all[: k + 1].copy()
    return ctx.d2h(pb, int(off[-1])), off, k


def cpu_baseline(ctx, pb, po, n_words, merges, budget_bytes, specials, max_seconds=20.0):
    """The reference's merge loop restated in pure Python with its own data structures (oracle/py_trainer.py: dict of pair
    counts, dict pair -> set of words, max over all pairs per iteration), ONE core -- the loop is single-threaded in the
    reference too (trainer.py:241-300) -- on the first `budget_bytes` of the same corpus, merge loop capped at max_seconds."""
    from oracle import py_trainer

    flat, off, k = _leading_words(ctx, pb, po, n_words, budget_bytes)
    fb, o = flat.tobytes(), off.tolist()
    t0 = time.time()
    words = [fb[o[i]:o[i + 1]] for i in range(k)]
    t_build = time.time() - t0
    t0 = time.time()
    _vocab, mg = py_trainer.merge_loop(words, 257 + merges, 1, specials, max_seconds=max_seconds)
    dt = time.time() - t0
    return {
        "value": round(len(mg) / dt, 3), "unit": "merges/s", "cores": 1, "kind": "port",
        "sample": f"first {int(off[-1])} bytes ({k} words) of the same corpus: {len(mg)} merges in {dt:.1f} s (word pooling and the initial pair "
                  f"count included, the loop stopped after {max_seconds:.0f} s; slicing the words took another {t_build:.1f} s); pure-Python "
                  f"restatement of the reference's trainer with its dict/set/max structures (oracle/py_trainer.py, pinned on the "
                  f"reference-made golden vectors); host has {os.cpu_count()} cores, the loop uses 1",
        "corpus_bytes_per_sec": round(int(off[-1]) / dt, 1),
    }


def cpu_baseline_c_port(ctx, pb, po, n_words, merges, budget_bytes, specials, max_seconds=10.0):
    """The C port of the same algorithm (oracle/bpe_oracle.c, the parity oracle), 1 core, same kind of bounded sample."""
    from oracle import oracle

    flat, off, k = _leading_words(ctx, pb, po, n_words, budget_bytes)
    t0 = time.time()
    _vocab, mg, info = oracle.train_flat(flat, off, 257 + merges, 1, specials, return_ids=True, max_seconds=max_seconds)
    dt = time.time() - t0
    return {"value": round(len(mg) / dt, 2), "unit": "merges/s", "cores": 1, "kind": "port (C)",
            "sample": f"first {int(off[-1])} bytes ({k} words, {info['unique_words']} unique): {len(mg)} merges in {dt:.1f} s (loop stopped after {max_seconds:.0f} s)"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--target-mib", type=int, default=1024)
    ap.add_argument("--merges", type=int, default=32000)
    ap.add_argument("--event-sample", type=int, default=0, help="time every Nth sparse-phase launch with HIP events too (0 = off: the phase is timed as a whole)")
    ap.add_argument("--event-sample-dense", type=int, default=2, help="... every Nth launch of the streaming phase (fused k_apply)")
    ap.add_argument("--cpu-sample-mib", type=int, default=64)
    ap.add_argument("--long-words", type=int, default=0, help="auxiliary line: the job on 256 MiB of the corpus with N words of 64..300 bytes added (the long-word path), next to the same job without them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dedup-line", action="store_true")
    ap.add_argument("--no-pretok-line", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"note: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    import torch

    # Rehearsal switch (not used by the driver): YABPE_BENCH_REHEARSAL=1 runs all ranks on GPU 0 with gloo and the
    # torch transport, to exercise the multi-rank bench flow on a 1-GPU box (RCCL refuses two ranks on one device).
    rehearsal = os.environ.get("YABPE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from yet_another_bpe import _native, synth

    specials = ["<|endoftext|>"]
    base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
    spec = synth.SynthSpec.config3(args.target_mib << 20)

    gen = _native.Context(local_rank)
    t0 = time.time()
    pb, po, n_words, n_bytes = gen.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    log(f"[rank {rank}] corpus on device: {n_bytes} bytes, {n_words} words ({time.time()-t0:.2f} s)")

    if world > 1:
        from yet_another_bpe import distributed as ydist

        runner = ydist.ShardedRunner(gen, pb, po, n_words, n_bytes, base, rank, world, local_rank,
                                     transport=("torch" if rehearsal else "rccl") + ("" if os.environ.get("YABPE_P2P") == "0" else "+p2p"))
        runner._context()  # rendezvous + RCCL communicator before the timed region
    else:
        runner = None

    def one_job(dedup: bool, event_sample: int, timed: bool = False):
        if runner is not None:
            return runner.run(args.merges, 1, dedup=dedup, event_sample=event_sample)
        with _native.Context(local_rank) as ctx:
            ctx.set_option("event_sample", event_sample)
            ctx.set_option("event_sample_dense", args.event_sample_dense if timed else 0)
            ctx.set_vocab(base)
            ctx.load_words_ptr(pb, po, n_words, dedup=dedup)
            left, right, merged, count = ctx.train(args.merges, 1)
            return {"n_merges": len(left), "stats": ctx.stats(), "left": left, "right": right, "merged": merged}

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # code-object load + allocator warm-up on a tiny job (not a step)
    with _native.Context(local_rank) as wctx:
        import numpy as np

        f, o = synth.generate(synth.SynthSpec(64 << 10, 1000, 5, b"abcdef", True))
        wctx.set_vocab(base)
        wctx.load_words(f, o)
        wctx.train(20, 1)

    for _ in range(args.warmup):
        one_job(False, 0)
    barrier()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res = one_job(False, args.event_sample, timed=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if rehearsal else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = res["stats"]
    n_merges = res["n_merges"]
    out = {
        "metric": "merges_per_sec", "value": round(args.steps * n_merges / elapsed, 2), "unit": "merges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * elapsed / args.steps, 2),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": f"{args.target_mib} MiB synthetic byte corpus (SURVEY 8d config 3: 256-value alphabet, 1M Zipf types, seed 3), "
                               f"{args.merges} merges, min_frequency 1, flat layout (every word occurrence resident), "
                               f"{'word-sharded over %d GPUs' % world if world > 1 else '1 GPU'}",
                   "corpus_bytes": n_bytes, "words": n_words, "merges": n_merges},
        "corpus_bytes_per_sec": round(args.steps * n_bytes / elapsed, 1),
        "device_ms": {"load_tiles_and_initial_count": round(st["load_ms"], 2), "merge_loop": round(st["train_ms"], 2)},
        "stream": {"tokens_initial": st["tokens_initial"], "tokens_final": st["tokens_now"], "n_tiles": st["n_tiles"],
                   "live_slots_final": st["live_slots"], "retiles": st["retiles"], "table_entries": st["table_entries"],
                   "table_capacity": st["table_capacity"], "table_rebuilds": st["table_rebuilds"], "long_words": st["n_long_words"]},
    }
    if world > 1:
        out["exchange"] = {"exchanges": st["exchanges"], "window_bytes_per_exchange": st["exchange_bytes"],  # ([header | records transmitted at most] x ranks: what an all-gather moves; a peer-to-peer push carries only the records produced)
                           "record_capacity_per_rank": st["exchange_cap_records"],
                           "buffer_growths": st["exchange_growths"], "max_records_of_a_rank_at_a_batch_end": st["exchange_max_records"],
                           "merges_per_exchange": round(n_merges / max(1, st["exchanges"]), 2),
                           "peer_to_peer": bool(st["exchange_p2p"]),
                           "exchange_launch_us_sampled": round(1e3 * st["exchange_ms_sampled"] / st["exchanges_sampled"], 2) if st["exchanges_sampled"] else None,
                           "note": "per batch of merges: apply launch (deltas leave as records) -> the exchange -> one launch that adds every rank's records "
                                   "to the replica and selects the next batch.  peer_to_peer: the exchange is ONE small launch that pushes this rank's "
                                   "records into the peers' hipIpc-mapped buffers (xGMI), raises a flag there and waits for theirs "
                                   "(exchange_launch_us_sampled: its device time, one per round of launches, waiting for the slowest peer included); "
                                   "otherwise an RCCL all-gather of [header | records] on the compute stream (YABPE_P2P=0, or a rank could not map a peer)"}
    # ---- what bounds the timed job.  It has two phases:
    #   streaming (the first few dozen merges, most tiles change): ONE fused launch per selection (a batch of up to two merges, in this
    #   corpus nearly always one), k_apply -- a coalesced read of the whole live token stream + rewrite + pair-table update + the selection of
    #   the next merges.  HBM-bound by design: `roofline` = the phase's A_i = 2 (T_i + W) per launch / the event-timed launches' average
    #   duration (whole iterations).
    #   sparse (the rest: a tile-level skip index finds the ~1 % of the tiles that hold a pair): ONE fused launch per BATCH of
    #   merges (the selection fixes up to 16 consecutive merges it can prove independent), k_scan_skip -- a chain of dependent
    #   memory trips per launch, not a streaming kernel.  `latency` = its measured time per launch against a floor built from
    #   pieces measured on this device in this run (yabpe_latency_probe), and the resulting time per merge.
    phase_ms = {"streaming": round(st["train_ms"] - st["sparse_ms"], 2), "sparse": round(st["sparse_ms"], 2)}
    out["device_ms"]["phases"] = phase_ms
    if st["dense_launches_sampled"]:
        n_l = st["dense_launches_sampled"]
        secs = st["dense_ms_sampled"] / 1000.0
        algo, actual = st["dense_algo_bytes_sampled"], st["dense_actual_bytes_sampled"]
        achieved = algo / secs / 1e9
        out["roofline"] = {
            "kernel": "k_apply<false,true> (fused launch of the streaming phase: one pass over the stream applies the merges the last selection "
                      "batched -- scan + rewrite + table update + selection)", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "launches_timed": n_l, "avg_launch_us": round(1e6 * secs / n_l, 2), "algo_bytes_per_launch": algo // n_l,
            "launches": st["dense_launches"], "merges": st["dense_merges"], "merges_per_launch": round(st["dense_merges"] / max(1, st["dense_launches"]), 3),
            "actual_stream_bytes_per_launch": actual // n_l, "actual_stream_GBps": round(actual / secs / 1e9, 1),
            "share_of_merge_loop_time": round(phase_ms["streaming"] / max(st["train_ms"], 1e-9), 4),
            "measured_on": f"the timed job itself: every {args.event_sample_dense}th launch of its streaming phase "
                           f"({st['merges_done'] - st['sparse_merges']} merges), HIP events on the library's stream",
            "note": "achieved = algorithmic bytes per launch / average launch duration: sum of 2*(T_i+W) over the merges the phase's launches applied "
                    "/ its launches (a launch applies merges_per_launch merges in one pass), over the average of the event-timed launches, whole iterations "
                    "(T_i live tokens, W words; SURVEY 8d); actual_* = the u16 slots really read (single-token words leave the stream).  traffic: PMC counters cannot be "
                    "collected inside this run; the rocprofv3 --pmc summary for this kernel is kept under profiles/ (see traffic_profile)",
        }
        pmc = sorted((REPO / "profiles").glob("r*_pmc_k_apply_summary.json"))
        if pmc and args.target_mib == 1024 and world == 1:
            d = json.loads(pmc[-1].read_text())
            tb = int(d["traffic_bytes_per_launch"])
            out["roofline"]["traffic"] = tb
            out["roofline"]["traffic_source"] = f"profiles/{pmc[-1].name}: a committed rocprofv3 --pmc profile of this kernel on this workload, NOT measured in this run"
            out["roofline"]["traffic_profile"] = {
                "file": f"profiles/{pmc[-1].name}", "traffic_bytes_per_launch": tb,
                "traffic_GBps_at_this_runs_launch_time": round(tb / (secs / n_l) / 1e9, 1), "traffic_frac_of_peak": round(tb / (secs / n_l) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "`traffic` comes from that committed rocprofv3 --pmc profile of the same kernel on the same workload (tools/collect_pmc.sh), NOT "
                        f"from this run.  It is {tb / max(1, algo // n_l):.2f} x the algorithmic bytes: the stream is read once and the rewrite writes back every "
                        "tile from its first changed slot on -- in the first merges nearly every tile changes, so the algorithmic figure "
                        "(reads of live tokens only) cannot be met by any in-place rewrite; against the bytes the kernel has to move the launch "
                        "runs at the traffic_frac_of_peak given here"}
    if st["sparse_merges"] and rank == 0:
        lat = gen.latency_probe()
        n_load, n_coh, n_atomic = 4, 4, 3
        floor = lat["launch_gap_us"] + n_load * lat["load_trip_us"] + n_coh * lat["coherent_trip_us"] + n_atomic * lat["atomic_trip_us"]
        launches = max(1, st["sparse_launches"])
        per_launch = 1000.0 * st["sparse_ms"] / launches
        per_merge = 1000.0 * st["sparse_ms"] / st["sparse_merges"]
        tail_launch = 1000.0 * st["tail_ms"] / max(1, st["tail_launches"])
        out["latency"] = {
            "kernel": "k_scan_skip (fused launch of the sparse phase: applies a batch of merges, selects the next batch)", "bound": "dependent memory trips per launch",
            "floor_us_per_launch": round(floor, 2), "achieved_us_per_launch": round(per_launch, 2), "frac": round(floor / per_launch, 4),
            "merges": st["sparse_merges"], "launches": st["sparse_launches"], "mean_batch": round(st["sparse_merges"] / launches, 2),
            "achieved_us_per_merge": round(per_merge, 2),
            "share_of_merge_loop_time": round(phase_ms["sparse"] / max(st["train_ms"], 1e-9), 4),
            "second_half_us_per_merge": round(1000.0 * st["tail_ms"] / max(1, st["tail_merges"]), 2),
            "second_half_us_per_launch": round(tail_launch, 2) if st["tail_launches"] else None,
            "second_half_mean_batch": round(st["tail_merges"] / max(1, st["tail_launches"]), 2) if st["tail_launches"] else None,
            "second_half_frac": round(floor / tail_launch, 4) if st["tail_launches"] else None,
            "pieces_us": {k: round(v, 3) for k, v in lat.items()},
            "model": f"floor of ONE launch = 1 launch boundary + {n_load} cache-missing loads (batch record, signature word, tile, byte-string set probe) + "
                     f"{n_coh} device-scope loads (table key, candidate list, counts, records of the window's pairs) + {n_atomic} returning atomics "
                     "(count add, two ticket levels): the dependent chain of a launch when every step takes one idle-device trip and compute, "
                     "bandwidth, imbalance and queueing take nothing -- whatever the number of merges it applies.  achieved = the whole sparse "
                     "phase (its first merges still have 10^5 sites each: those launches are bound by the tiles they read, not by trips) "
                     "including the host's housekeeping between rounds of launches (candidate-list and signature rebuilds, table growth, "
                     "retile); second_half_* = the same for the last half of the merges, where a merge has a few thousand sites",
        }
        if st["apply_launches_sampled"] and st["scan_launches_sampled"]:
            n_l = st["scan_launches_sampled"]
            tiles_read = st["scan_skip_tiles_read"] / max(1, st["scan_skip_launches"])
            out["latency"]["event_timed"] = {"launches": n_l, "avg_launch_us": round(1e3 * st["scan_ms_sampled"] / max(1, n_l), 2),
                                             "avg_fraction_of_tiles_read": round(tiles_read / max(1, st["n_tiles"]), 4)}
    if rank == 0 and not args.no_dedup_line and world == 1:
        t1 = time.perf_counter()
        rd = one_job(True, 0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        same = bool((rd["left"] == res["left"]).all() and (rd["right"] == res["right"]).all()) if rd["n_merges"] == n_merges else False
        out["dedup_layout"] = {"merges_per_sec": round(rd["n_merges"] / dt, 2), "seconds": round(dt, 3),
                               "unique_words": rd["stats"]["n_words"], "same_merges_as_flat": same,
                               "device_ms": {"dedup_load_count": round(rd["stats"]["load_ms"], 2), "merge_loop": round(rd["stats"]["train_ms"], 2)}}
    if rank == 0 and not args.no_pretok_line and world == 1:
        # the step before the path: GPT-2 pre-tokenisation of text resident in HBM.  Text = the config-2
        # style generator at this run's size (space + lower-case word, Zipf over 50k types), pre-tokens must equal its words.
        with _native.Context(local_rank) as pt:
            tb, _to, tw, tn = pt.synth_generate(args.target_mib << 20, 50_000, 2, b"abcdefghijklmnopqrstuvwxyz", True)
            best = None
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                _dt, _do, nw_pt = pt.pretokenize(tb, n_bytes=tn, special_tokens=["<|endoftext|>"])
                dt = time.perf_counter() - t1
                pt.pretokenize_free()
                best = dt if best is None else min(best, dt)
            cpu_pt = None
            if not args.no_cpu_baseline:  # what the reference runs for this step: regex.findall, one core, on a bounded sample
                from oracle import pretok

                sample = pt.d2h(tb, min(tn, 16 << 20)).tobytes()
                sample = sample[: sample.rfind(b" ")]
                t1 = time.perf_counter()
                n_cpu = len(pretok.pretokenize(sample, ["<|endoftext|>"]))
                dt = time.perf_counter() - t1
                cpu_pt = {"MB_per_sec": round(len(sample) / dt / 1e6, 2), "cores": 1, "kind": "reference dependency (regex.findall)",
                          "sample": f"first {len(sample)} bytes of the same text, {n_cpu} pre-tokens in {dt:.2f} s"}
            import numpy as np

            _dt, do2, nw2 = pt.pretokenize(tb, n_bytes=tn, special_tokens=["<|endoftext|>"])
            same_offsets = bool(nw2 == tw and np.array_equal(pt.d2h(do2, (nw2 + 1) * 8, dtype=np.uint64), pt.d2h(_to, (tw + 1) * 8, dtype=np.uint64)))
            pt.pretokenize_free()
            out["pretokenize"] = {"GB_per_sec": round(tn / best / 1e9, 2), "ms": round(best * 1e3, 2), "text_bytes": tn, "pretokens": nw_pt,
                                  "offsets_equal_generator_words": same_offsets, "cpu_baseline": cpu_pt,
                                  "note": "yabpe_pretokenize (UTF-8 validation + GPT-2 split + special tokens -> word offsets in HBM), "
                                          "wall time incl. scratch allocation, best of 3; not part of `value`"}
    if rank == 0 and args.long_words > 0 and world == 1:
        # the long-word path (words of more than 63 tokens live outside the tile stream): what it costs per merge, measured
        import numpy as np

        flat, off, k = _leading_words(gen, pb, po, n_words, 256 << 20)
        rng = np.random.default_rng(7)
        ll = rng.integers(64, 301, size=args.long_words).astype(np.uint64)
        lflat = rng.choice(np.frombuffer(b"etaoinshrdlu", dtype=np.uint8), size=int(ll.sum())).astype(np.uint8)
        off2 = np.concatenate([off, off[-1] + np.cumsum(ll)]).astype(np.uint64)
        flat2 = np.concatenate([flat, lflat])
        res_lw = {}
        for name, (f_, o_) in {"without": (flat, off), "with": (flat2, off2)}.items():
            with _native.Context(local_rank) as ctx:
                ctx.set_vocab(base)
                ctx.load_words(f_, o_)
                l_, _r, _m, _c = ctx.train(min(args.merges, 8000), 1)
                s_ = ctx.stats()
                res_lw[name] = {"merges": len(l_), "merge_loop_ms": round(s_["train_ms"], 2), "us_per_merge": round(1000.0 * s_["train_ms"] / max(1, len(l_)), 2),
                                "long_words": s_["n_long_words"]}
        out["long_words"] = {"corpus": f"first {int(off[-1])} bytes of the corpus, + {args.long_words} words of 64..300 random letters", **res_lw}
    if rank == 0 and not args.no_cpu_baseline and world == 1:  # (the CPU baseline is an N=1 item)
        out["cpu_baseline"] = cpu_baseline(gen, pb, po, n_words, args.merges, args.cpu_sample_mib << 20, specials)
        out["cpu_baseline_c_port"] = cpu_baseline_c_port(gen, pb, po, n_words, args.merges, 32 << 20, specials)
    if runner is not None:
        runner.close()
    gen.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
