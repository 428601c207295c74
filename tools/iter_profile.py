#!/usr/bin/env python3
"""Dev tool: per-iteration (live bytes, sites, k_apply microseconds) for a config-3 style run.
   python tools/iter_profile.py --target-mib 256 --merges 3000 --out gpurun_out/iter.csv"""
import argparse, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
import numpy as np
from yet_another_bpe import _native, synth

ap = argparse.ArgumentParser()
ap.add_argument("--target-mib", type=int, default=256)
ap.add_argument("--merges", type=int, default=3000)
ap.add_argument("--sample", type=int, default=1)
ap.add_argument("--out", default="gpurun_out/iter.csv")
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
spec = synth.SynthSpec.config3(a.target_mib << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    with _native.Context() as ctx:
        ctx.set_option("event_sample", a.sample)
        for kv in a.opt:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        ctx.set_vocab(base)
        ctx.load_words_ptr(pb, po, nw)
        left, right, merged, count = ctx.train(a.merges, 1)
        sites, live = ctx.iter_log()
        it, us, scan = ctx.event_log()
        st = ctx.stats()
Path(a.out).parent.mkdir(parents=True, exist_ok=True)
with open(a.out, "w") as f:
    f.write("iter,live_slots,sites,count,apply_us,scan_us,left_len,right_len\n")
    tl = [1] * 70000
    for k, (l, r, m) in enumerate(zip(left.tolist(), right.tolist(), merged.tolist())):
        if m < len(tl) and m >= 257:
            tl[m] = tl[l] + tl[r]
    for i, u, sc in zip(it.tolist(), us.tolist(), scan.tolist()):
        f.write(f"{i},{live[i]},{sites[i]},{count[i]},{u:.2f},{sc:.2f},{tl[left[i]]},{tl[right[i]]}\n")
print("skip launches", st["scan_skip_launches"], "tiles read", st["scan_skip_tiles_read"], "avg pass", st["scan_skip_tiles_read"] / max(1, st["scan_skip_launches"]) / max(1, st["n_tiles"]))
import collections
late = [(tl[left[i]] == 1 and tl[right[i]] == 1, sc, u) for i, u, sc in zip(it.tolist(), us.tolist(), scan.tolist()) if i >= 1000]
for flag in (True, False):
    xs = [x for x in late if x[0] == flag]
    if xs:
        print("iters>=1000", "byte-byte" if flag else "other", len(xs), "scan avg us", sum(x[1] for x in xs) / len(xs), "apply avg us", sum(x[2] for x in xs) / len(xs))
print("train_ms", st["train_ms"], "apply_ms_sampled", st["apply_ms_sampled"], "scan_ms_sampled", st["scan_ms_sampled"], "scan_launches", st["scan_launches_sampled"], "n", len(it), "retiles", st["retiles"], "rebuilds", st["table_rebuilds"], "table_cap", st["table_capacity"], "entries", st["table_entries"])
sel = [0, 1, 2, 5, 10, 20, 50, 100, 200, 400, 800, 1200, 1600, 2000, 2400, 2800, 2999, 5000, 10000, 20000, 31999]
for i in sel:
    if i < len(us):
        k = it[i]
        print(f"iter {k:6d} live {live[k]/1e6:8.1f}M slots  sites {sites[k]:9d}  apply {us[i]:8.1f} us  {2*live[k]/us[i]/1e3:7.1f} GB/s")
