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
        sh = np.zeros(65536, dtype=np.uint64)
        L.yabpe_debug_stop_hist(ctypes.c_void_p(sh.ctypes.data))
raw = out.reshape(65536, 4)
names = {0: "-", 1: "batch rule (1) / (2) or a count below the threshold", 4: "limit", 5: "window end", 6: "run merge"}
for lo, hi in [(80, 300), (300, 1000), (1000, 3000), (3000, 12000), (12000, 32000)]:
    v = sh[lo:hi]; v = v[v != 0]
    if len(v) == 0: continue
    why = (v & np.uint64(0xff)).astype(int); n = ((v >> np.uint64(8)) & np.uint64(0xff)).astype(int); nw = (v >> np.uint64(16)).astype(int)
    print(f"selections at merges {lo}-{hi}: {len(v)}  mean batch {n.mean():.2f}  window entries mean {nw.mean():.1f} max {nw.max()} | ended by: " + ", ".join(f"{names.get(k, k)} {np.mean(why == k) * 100:.0f}%" for k in sorted(set(why))))
p = raw.astype(np.float64) / 100.0  # us
valid = np.nonzero((raw[:, 0] != np.uint64(0xFFFFFFFFFFFFFFFF)) & (raw[:, 1] != 0) & (raw[:, 3] != 0))[0]  # row = DevState::iter when the launch started
nxt = valid[1:]
cur = valid[:-1]
batch = nxt - cur                      # merges the launch's selection added = what the NEXT launch applies
wg = p[cur, 1] - p[cur, 0]             # first workgroup start -> last workgroup end (apply + flush + ticket wait of the others)
sel = p[cur, 3] - p[cur, 1]            # -> DevState of the next batch stored
gap = p[nxt, 0] - p[cur, 3]            # -> first workgroup of the next launch
per = p[nxt, 0] - p[cur, 0]
applied = np.concatenate([[1], batch[:-1]])  # merges this launch applied (selected by the launch before it)
good = (gap > 0) & (gap < 500) & (per < 5000)
for lo, hi in [(100, 300), (300, 1000), (1000, 3000), (3000, 8000), (8000, 12000), (12000, 20000), (20000, 31990)]:
    m = good & (cur >= lo) & (cur < hi)
    if not m.any():
        continue
    print(f"merges {lo:6d}-{hi:6d}: launches {m.sum():5d} mean batch {applied[m].mean():.2f} | medians: workgroups {np.median(wg[m]):7.2f}  selection tail {np.median(sel[m]):6.2f}  "
          f"between launches {np.median(gap[m]):5.2f}  period {np.median(per[m]):7.2f} us | mean period {per[m].mean():7.2f} us = {per[m].sum() / applied[m].sum():6.2f} us per merge")
    for k in (1, 2, 3, 4, 6, 8):
        mk = m & (applied == k)
        ms = m & (batch == k)
        if mk.sum() >= 5:
            print(f"      launches applying {k} merges: n {mk.sum():5d}  workgroups {np.median(wg[mk]):7.2f} (mean {wg[mk].mean():7.2f})   |  selecting {k}: n {ms.sum():5d} selection tail {np.median(sel[ms]) if ms.any() else 0:6.2f}")
