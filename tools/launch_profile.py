#!/usr/bin/env python3
"""Dev tool (needs libyabpe_launchprof.so: make -C yet-another-bpe_amd/csrc libyabpe_launchprof.so): per-merge timeline of the fused launches from device wall-clock stamps --
first workgroup start, last workgroup end, selection end -- and from them the time BETWEEN launches (end of the
selection of launch i -> first workgroup of launch i+1), which no in-kernel stamp and no rocprof duration shows alone."""
import ctypes, os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
os.environ["YABPE_LIB"] = str(REPO / "yet-another-bpe_amd/csrc/libyabpe_launchprof.so")
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
import numpy as np
from yet_another_bpe import _native, synth
spec = synth.SynthSpec.config3(1024 << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
opts = [kv.split("=") for kv in sys.argv[1:]]
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    with _native.Context() as ctx:
        for k, v in opts:
            ctx.set_option(k, int(v))
        ctx.set_vocab(base); ctx.load_words_ptr(pb, po, nw)
        L = _native.lib()
        L.yabpe_debug_launch_profile(None, 1)
        ctx.train(32000, 1)
        sp = (ctypes.c_uint64 * 16)(); L.yabpe_debug_sel_profile(sp); sp = [int(v) for v in sp]
        rel = lambda i: (sp[i] - sp[0]) / 100.0
        print("last launch, selection tail (us after its workgroup reached the ticket): ticket won %.2f | entries+length in %.2f | counts in %.2f | tie records in %.2f | state+counters folded %.2f | winner %.2f | records %.2f | probe %.2f | compare %.2f | end %.2f"
              % (rel(8), rel(9), rel(10), rel(11), rel(2), rel(3), rel(4), rel(5), rel(6), rel(7)))
        out = np.zeros(65536 * 4, dtype=np.uint64)
        L.yabpe_debug_launch_profile(ctypes.c_void_p(out.ctypes.data), 0)
raw = out.reshape(65536, 4)
print("raw rows 25000..25003:", raw[25000:25004].tolist())
p = raw.astype(np.float64) / 100.0  # us
ok = (out.reshape(65536, 4)[:, 0] != np.uint64(0xFFFFFFFFFFFFFFFF)) & (out.reshape(65536, 4)[:, 1] != 0)
for lo, hi in [(300, 1000), (1000, 3000), (3000, 8000), (8000, 12000), (12000, 20000), (20000, 31990)]:
    idx = np.arange(lo, hi)
    idx = idx[ok[idx] & ok[idx + 1] & (p[idx, 3] > 0)]
    wg = p[idx, 1] - p[idx, 0]              # first workgroup start -> last workgroup end (apply + flush + ticket wait of the others)
    sel = p[idx, 3] - p[idx, 1]             # -> DevState of the next merge stored
    gap = p[idx + 1, 0] - p[idx, 3]         # -> first workgroup of the next launch
    per = p[idx + 1, 0] - p[idx, 0]
    good = (gap > 0) & (gap < 200) & (per < 1000)
    print(f"merges {lo:6d}-{hi:6d} n={good.sum():6d}: workgroups {np.median(wg[good]):6.2f}  selection tail {np.median(sel[good]):6.2f}  "
          f"between launches {np.median(gap[good]):6.2f}  period {np.median(per[good]):6.2f} us (medians; mean period {per[good].mean():.2f})")
