#!/usr/bin/env python3
"""Dev tool (needs libyabpe_scanprof.so: make -C yet-another-bpe_amd/csrc libyabpe_scanprof.so): where the time of the fused
batch selection goes, averaged over the selections of the 1 GiB / 32,000-merge job by phase of the job.  The stamps are
the 100 MHz wall clock of the selecting workgroup's thread 0, relative to the moment it won the ticket."""
import ctypes, os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
os.environ["YABPE_LIB"] = os.environ.get("SCANPROF_LIB", str(REPO / "yet-another-bpe_amd/csrc/libyabpe_scanprof.so"))
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
from yet_another_bpe import _native, synth
spec = synth.SynthSpec.config3(1024 << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    with _native.Context() as ctx:
        for kv in sys.argv[1:]:
            k, v = kv.split("="); ctx.set_option(k, int(v))
        ctx.set_vocab(base); ctx.load_words_ptr(pb, po, nw)
        ctx.train(32000, 1)
        acc = (ctypes.c_uint64 * 72)(); _native.lib().yabpe_debug_sel_acc(acc); acc = [int(v) for v in acc]
names = [(1, "selection starts"), (9, "round trips 1 + 2 in (list entries, state, counters; counts)"), (10, "maximum of this thread's counts"), (12, "maximum over the workgroup"),
         (13, "window built (distinct pairs near the maximum)"), (11, "round trip 3 in (token records)"), (14, "order and batch rule, all entries side by side"),
         (3, "state folded, stop rules"), (5, "batch cut, round trip 4 in (byte-string set)"), (7, "committed")]
for r, title in enumerate(["merges < 3,000", "merges 3,000-12,000", "merges >= 12,000"]):
    a = acc[24 * r:24 * r + 24]
    if not a[0]: continue
    n = a[0]
    print(f"{title}: {n} selections, mean batch {a[16] / n:.2f}, window entries {a[17] / n:.1f}, first workgroup at the tail -> ticket won {a[18] / n / 100:.2f} us")
    prev = 0.0
    for i, nm in names:
        t = a[i] / n / 100.0
        print(f"    +{t - prev:5.2f} us  (at {t:5.2f})  {nm}")
        prev = t
