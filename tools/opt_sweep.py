#!/usr/bin/env python3
"""Dev tool: the config-3 job (1 GiB, 32,000 merges) under different tunables; prints the device time of the merge loop.
   python tools/opt_sweep.py "" "full_skip=0" "full_skip_multi=4096" "scan_skip_blocks=768,full_skip_blocks=512" """
import sys, hashlib
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
from yet_another_bpe import _native, synth
import os
mib = int(os.environ.get("SWEEP_MIB", "1024")); merges = int(os.environ.get("SWEEP_MERGES", "32000"))
spec = synth.SynthSpec.config3(mib << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
ref = None
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    for cfg in (sys.argv[1:] or [""]):
        with _native.Context() as ctx:
            for kv in filter(None, cfg.split(",")):
                k, v = kv.split("=")
                ctx.set_option(k, int(v))
            ctx.set_vocab(base)
            ctx.load_words_ptr(pb, po, nw)
            left, right, merged, count = ctx.train(merges, 1)
            st = ctx.stats()
            h = hashlib.sha256(left.tobytes() + right.tobytes() + merged.tobytes()).hexdigest()[:12]
            ref = ref or h
            print(f"{cfg or '(defaults)':60s} train {st['train_ms']:8.1f} ms  {len(left) / st['train_ms'] * 1e3:8.0f} merges/s  merges {h} {'==' if h == ref else '!= FIRST'}", flush=True)
