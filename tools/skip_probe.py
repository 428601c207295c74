#!/usr/bin/env python3
"""Dev tool: how selective the skip index is over a job -- tiles read per merge next to the merge's sites, by segment.
   python tools/skip_probe.py [--mib M] [--opt k=v ...]"""
import argparse, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
import numpy as np
from yet_another_bpe import _native, synth
ap = argparse.ArgumentParser()
ap.add_argument("--mib", type=int, default=1024)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
spec = synth.SynthSpec.config3(a.mib << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
edges = [300, 1000, 3000, 8000, 12000, 16000, 20000, 24000, 28000, 32000]
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    with _native.Context() as ctx:
        for kv in a.opt:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        ctx.set_vocab(base)
        ctx.load_words_ptr(pb, po, nw)
        done, prev_read, prev_launch, prev_ms = 0, 0, 0, 0.0
        for e in edges:
            ctx.train(e - done, 1)
            st = ctx.stats()
            sites, live = ctx.iter_log()
            s = np.asarray(sites[done:e], dtype=np.float64)
            rd, ln = st["scan_skip_tiles_read"] - prev_read, st["scan_skip_launches"] - prev_launch
            print(f"merges {done:6d}-{e:6d}: tiles {st['n_tiles']}  read/merge {rd/max(ln,1):9.0f}  sites/merge mean {s.mean():9.0f} median {np.median(s):8.0f}"
                  f"  read/sites {rd/max(s.sum(),1):6.2f}  ms {st['train_ms']-prev_ms:7.1f}  sig_builds {st.get('sig_builds','?')} retiles {st['retiles']}", flush=True)
            done, prev_read, prev_launch, prev_ms = e, st["scan_skip_tiles_read"], st["scan_skip_launches"], st["train_ms"]
