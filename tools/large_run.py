#!/usr/bin/env python3
"""Dev tool: one GPU, a corpus sized for its 288 GB -- N GiB of synthetic text (config-2 style generator) resident in HBM,
pre-tokenised on the device, trained in the FLAT layout (every occurrence resident: ~2.3 bytes of tile + 4.7 bytes of
signatures per corpus byte) and, for comparison, with pooled words.  Checks that the two layouts give the same merges and
that the incrementally maintained table equals a recount of the final stream.
    python tools/large_run.py [GiB=8] [merges=2000]"""
import hashlib, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
import numpy as np
from yet_another_bpe import _native

gib = int(sys.argv[1]) if len(sys.argv) > 1 else 8
merges = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
SP = ["<|endoftext|>"]
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
with _native.Context() as gen:
    t0 = time.perf_counter()
    tb, to, tw, tn = gen.synth_generate(gib << 30, 50_000, 2, b"abcdefghijklmnopqrstuvwxyz", True)
    print(f"text: {tn / 2**30:.2f} GiB, {tw} words generated on the device in {time.perf_counter() - t0:.2f} s", flush=True)
    t0 = time.perf_counter()
    dt, do, nw = gen.pretokenize(tb, n_bytes=tn, special_tokens=SP)
    print(f"pre-tokenised: {nw} pre-tokens in {(time.perf_counter() - t0) * 1e3:.0f} ms ({tn / (time.perf_counter() - t0) / 1e9:.1f} GB/s)", flush=True)
    assert nw == tw
    digests = []
    for dedup in (True, False):
        with _native.Context() as ctx:
            ctx.set_vocab(base)
            t0 = time.perf_counter()
            ctx.load_words_ptr(dt, do, nw, dedup=dedup)
            t_load = time.perf_counter() - t0
            t0 = time.perf_counter()
            left, right, merged, count = ctx.train(merges, 1)
            t_train = time.perf_counter() - t0
            st = ctx.stats()
            mm = ctx.verify_table()
            d = hashlib.sha256(left.tobytes() + right.tobytes() + merged.tobytes()).hexdigest()[:16]
            digests.append(d)
            print(f"{'pooled' if dedup else 'flat  '}: words resident {st['n_words']}, tiles {st['n_tiles']}, load {t_load:.2f} s, {len(left)} merges in {t_train:.2f} s "
                  f"({len(left) / t_train:.0f} merges/s), retiles {st['retiles']}, table mismatches {mm}, merges {d}", flush=True)
            assert mm == 0
    assert digests[0] == digests[1], "layouts disagree"
    print("ok: flat and pooled layouts agree")
