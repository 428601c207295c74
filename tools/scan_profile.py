#!/usr/bin/env python3
"""Dev tool: timeline of one k_scan_skip launch (needs libyabpe_scanprof.so: make -C yet-another-bpe_amd/csrc libyabpe_scanprof.so).
Every workgroup stamps the 100 MHz wall clock at: 0 start, 1 signatures tested, 2 candidates matched, 3 single-site
rewrites done, 4 loop left, 7 end (after the aggregator flush).  Prints the spread of each phase over the workgroups."""
import ctypes, os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
os.environ["YABPE_LIB"] = os.environ.get("SCANPROF_LIB", str(REPO / "yet-another-bpe_amd/csrc/libyabpe_scanprof.so"))
sys.path.insert(0, str(REPO / "yet-another-bpe_amd"))
import numpy as np
from yet_another_bpe import _native, synth
merges = [int(x) for x in (sys.argv[1:] or ["3000", "20000"])]
WPB = int(os.environ.get("WPB", "8"))  # waves per workgroup of the sparse launch (forced through the option full_wpb)
spec = synth.SynthSpec.config3(1024 << 20)
base = [bytes([b]) for b in range(256)] + [b"<|endoftext|>"]
NB = 4096
with _native.Context() as g:
    pb, po, nw, nb = g.synth_generate(spec.target_bytes, spec.n_types, spec.seed, spec.alphabet, spec.space_prefix)
    with _native.Context() as ctx:
        ctx.set_option("full_wpb", WPB)
        if os.environ.get("BATCH_MAX"): ctx.set_option("batch_max", int(os.environ["BATCH_MAX"]))
        ctx.set_vocab(base); ctx.load_words_ptr(pb, po, nw, dedup=bool(int(os.environ.get("DEDUP", "0"))))  # DEDUP=1: the pooled layout
        done = 0
        for m in merges:
            ctx.train(m - done, 1); done = m
            st = ctx.stats()
            out = np.zeros(NB * 8, dtype=np.uint64)
            _native.lib().yabpe_debug_scan_profile(ctypes.c_void_p(out.ctypes.data), ctypes.c_uint32(NB))
            p = out.reshape(NB, 8).astype(np.int64)
            nt = 64 * WPB
            target = 256 * 16 // WPB  # (full_skip_blocks: 16 waves per CU)
            chunk = min(nt * (2 if WPB >= 16 else 4), max(64, (-(-st["n_tiles"] // target) + 63) // 64 * 64))  # as the host sizes it
            n_scan = min(-(-st["n_tiles"] // chunk), target)
            recent = p[:, 7].max() - 30000  # stamps older than 300 us belong to earlier launches
            scan = p[:n_scan]; rank = p[n_scan:]; rank = rank[rank[:, 0] > recent]
            scan = scan[scan[:, 0] > recent]
            t0 = min(scan[:, 0].min(), rank[:, 0].min() if len(rank) else scan[:, 0].min())
            us = lambda x: x / 100.0
            kk = scan[:, 3] >> 32; ncand = scan[:, 3] & 0xffffffff
            print(f"--- after {m} merges: n_tiles {st['n_tiles']} scan blocks {n_scan} rank blocks {len(rank)} | merges in the profiled launch {int(np.median(kk))}, candidate tiles per workgroup mean {ncand.mean():.1f} max {ncand.max()} (per wave {ncand.mean() / WPB:.2f})")
            print(f"span scan blocks: first start 0, last start {us(scan[:,0].max()-t0):.2f}, last end {us(scan[:,7].max()-t0):.2f} us")
            if len(rank):
                print(f"rank blocks: start {us(rank[:,0].min()-t0):.2f}..{us(rank[:,0].max()-t0):.2f}, end max {us(rank[:,7].max()-t0):.2f}, dur mean {us((rank[:,7]-rank[:,0]).mean()):.2f} max {us((rank[:,7]-rank[:,0]).max()):.2f}")
            names = ["merge->LDS initialised", "LDS init->sig tested", "sig->candidates matched", "cand->loop left", "flush (epilogue)"]
            idx = [(5, 6), (6, 1), (1, 2), (2, 4), (4, 7)]
            for nm, (a, b) in zip(names, idx):
                d = us(scan[:, b] - scan[:, a])
                print(f"  {nm:28s} mean {d.mean():7.2f}  p50 {np.percentile(d,50):7.2f}  p99 {np.percentile(d,99):7.2f}  max {d.max():7.2f} us")
            w0 = []
            if len(w0):
                for nm, (a, b) in (("wave 0: reload arrives", (2, 5)), ("wave 0: first rewrite", (5, 6))):
                    d = us(w0[:, b] - w0[:, a])
                    print(f"  {nm:28s} mean {d.mean():7.2f}  p50 {np.percentile(d,50):7.2f}  p99 {np.percentile(d,99):7.2f}  max {d.max():7.2f} us  (n={len(w0)})")
            d = us(scan[:, 7] - scan[:, 0])
            print(f"  {'whole block':28s} mean {d.mean():7.2f}  p50 {np.percentile(d,50):7.2f}  p99 {np.percentile(d,99):7.2f}  max {d.max():7.2f} us")
            fo = np.zeros(NB * 4, dtype=np.uint64)
            _native.lib().yabpe_debug_flush_profile(ctypes.c_void_p(fo.ctypes.data), ctypes.c_uint32(NB))
            f = fo.reshape(NB, 4).astype(np.int64)[:n_scan]
            f = f[p[:n_scan][:, 0] > recent]
            for nm, d in (("flush: earlier stores acked", us(f[:, 0] - scan[:, 4])), ("flush: table keys arrived", us(f[:, 1] - f[:, 0])),
                          ("flush: count adds returned", us(f[:, 2] - f[:, 1])), ("flush: rest", us(scan[:, 7] - f[:, 2]))):
                print(f"  {nm:28s} mean {d.mean():7.2f}  p50 {np.percentile(d,50):7.2f}  p99 {np.percentile(d,99):7.2f}  max {d.max():7.2f} us")
            late = np.argsort(scan[:, 7])[-max(8, len(scan) // 20):]  # the workgroups that finish last: what the launch waits for
            print(f"  the last {len(late)} workgroups to finish: start->sig {us(scan[late,1]-scan[late,0]).mean():.2f}  candidates {us(scan[late,2]-scan[late,1]).mean():.2f}"
                  f"  flush: stores acked {us(f[late,0]-scan[late,4]).mean():.2f} keys {us(f[late,1]-f[late,0]).mean():.2f} adds {us(f[late,2]-f[late,1]).mean():.2f} rest {us(scan[late,7]-f[late,2]).mean():.2f}"
                  f"  | end at {us(scan[late,7]-t0).mean():.2f} (all: {us(scan[:,7]-t0).mean():.2f}) entries {f[late,3].mean():.1f}")
            ne = f[:, 3]
            print(f"  entries flushed per workgroup: mean {ne.mean():.1f} p99 {np.percentile(ne,99):.0f} max {ne.max()}; tiles read/WG n/a")
            for lo, hi in ((0, 1), (1, 8), (8, 16), (16, 32), (32, 64), (64, 10**9)):
                m_ = (ne >= lo) & (ne < hi)
                if m_.any():
                    print(f"    WGs with {lo:3d}..{hi if hi < 10**9 else 'inf'} entries: n {m_.sum():5d}  keys {us(f[m_,1]-f[m_,0]).mean():6.2f}  adds {us(f[m_,2]-f[m_,1]).mean():6.2f} (max {us(f[m_,2]-f[m_,1]).max():6.2f})  cand phase {us(scan[m_,2]-scan[m_,1]).mean():6.2f} us")
            ss = (ctypes.c_uint64 * 16)(); _native.lib().yabpe_debug_ss_profile(ss); ss = list(ss)
            if ss[9]:
                print("  candidate loop (workgroup 7, cumulative over the run): %d tiles, %.0f cycles per tile in the match+rewrite, %.1f %% rewritten in registers, %.1f %% by the general rewrite, %.1f %% without a site; %d groups, %.0f cycles waiting per group"
                      % (ss[9], ss[8] / ss[9], 100.0 * ss[10] / ss[9], 100.0 * ss[11] / ss[9], 100.0 * ss[12] / ss[9], ss[14], ss[13] / max(1, ss[14])))
            if ss[0]:
                print("  single_site_tile cycles/tile (workgroup 7, cumulative over the run): neighbours %.0f, deltas->LDS %.0f, sig bits %.0f, compaction+stores %.0f  (n=%d)"
                      % (ss[1] / ss[0], ss[2] / ss[0], ss[3] / ss[0], ss[4] / ss[0], ss[0]))
                if ss[6]:
                    print("  ... and then until the tile's stores and signature atomics are acknowledged: %.0f cycles (n=%d)" % (ss[5] / ss[6], ss[6]))
            sp = (ctypes.c_uint64 * 16)(); _native.lib().yabpe_debug_sel_profile(sp); sp = [int(v) for v in sp]
            rel = lambda i: (sp[i] - sp[0]) / 100.0
            print("  k_argmax_cand wg0: state loaded %.2f | list evaluated %.2f | bitmap evaluated %.2f us" % (rel(10), rel(11), rel(12)))
            print("  k_argmax_cand+select (last launch): wg0 partial stored %.2f | last ticket %.2f | loads in %.2f | winner decided %.2f | len/off %.2f | bytes+hash %.2f | probe %.2f | end %.2f us"
                  % (rel(9), rel(1), rel(2), rel(3), rel(4), rel(5), rel(6), rel(7)))
            for k in (1, 2, 3, 4, 7):
                print(f"  stamp {k}: min {us(scan[:,k].min()-t0):7.2f} max {us(scan[:,k].max()-t0):7.2f}")
