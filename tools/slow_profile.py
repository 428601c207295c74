#!/usr/bin/env python3
"""Dev tool: phase cycle breakdown of slow_tile (needs libyabpe_prof.so built with -DYB_PROFILE_SLOW)."""
import ctypes, os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
os.environ["YABPE_LIB"] = str(REPO / "yet-another-bpe_amd/csrc/libyabpe_prof.so")
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
from yet_another_bpe import _native, synth
spec = synth.SynthSpec.config3(1024 << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    with _native.Context() as ctx:
        ctx.set_vocab(base); ctx.load_words_ptr(pb, po, nw)
        m0, m1 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1000, 4000)  # merges [m0, m1) are profiled
        out = (ctypes.c_uint64 * 8)()
        if m0:
            ctx.train(m0, 1)
        _native.lib().yabpe_debug_slow_profile(out); a = list(out)
        ctx.train(m1 - m0, 1)
        _native.lib().yabpe_debug_slow_profile(out); b = list(out)
d = [y - x for x, y in zip(a, b)]
print("tiles", d[0])
names = ["", "stage+masks", "site loop (deltas, sig, agg)", "prefix+scatter", "write-back"]
for i in range(1, 5):
    print(f"{names[i]:32s} {d[i]/max(d[0],1):10.0f} cycles/tile")
print("total cycles/tile", sum(d[1:5]) / max(d[0], 1))
print(f"between two rewrites (wait for the next tile + match) {d[5]/max(d[0],1):10.0f} cycles/tile")
