// yabpe_kernels.h -- HIP kernels of the BPE training hot path for gfx950 (MI355X, CDNA4).
//
// One translation unit (yabpe.hip) includes this header.  DESIGN.md describes the data layout and each
// kernel's roofline; the short version:
//
//   token stream  u16 ids in tiles of YB_CAP slots (2 KiB); one wave owns one tile at a time; words never
//                 straddle tiles, every word ends with SEP, PAD fills lead/tail.  tile_len[t] = live slots.
//   pair table    open-addressing hash in HBM: keys u32 (left<<16|right), counts i64, updated with
//                 device-scope atomics; per-workgroup LDS hash aggregates deltas first.
//   k_apply       streaming phase: one coalesced read of the live token stream per merge (16 B per lane), match (a,b) on
//                 packed dwords, tiles with sites rewritten + compacted in place, pair-count deltas (tile_logic.h).
//   k_scan_skip   sparse phase: a tile-level skip index (blocked Bloom signatures) finds the ~1 % of the tiles that may hold
//                 the pair; only those are read and rewritten.
//   select_body   the next merge: exact argmax over a candidate list by (count, lexrank[left], lexrank[right])
//                 (trainer.py:246), stop rules, merged-token creation / byte-string identity (trainer.py:241-251, 296-300);
//                 run by the LAST workgroup of the launch that applied the previous merge (fused_select_tail).
//   rank blocks   extra workgroups of every launch keep lexrank[] = rank of every token's bytes in Python bytes order
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "tile_logic.h"

namespace yb {

constexpr int CAP = 1024;            // u16 slots per tile (2 KiB = 2 x 16 B per lane of a wave)
constexpr int LMAX = 64;             // longest word (tokens + SEP) kept in the tile stream
constexpr int SPAN = CAP - LMAX + 1; // packed positions owned by one tile when (re)tiling
constexpr int WPB = 4;               // waves per workgroup
constexpr int BLOCK = WPB * 64;
constexpr int AGG_N = 1024;          // LDS delta-aggregator entries per workgroup
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t PADPAD = (YB_PAD << 16) | YB_PAD;
constexpr int LONG_CH = 1024;        // long-word path: tokens per LDS chunk (the block rides on the per-batch launches: its LDS counts against theirs)
constexpr int SIG_ROWS = 512;        // tile signature: a blocked Bloom filter of the adjacent pairs present in the tile --
constexpr int SIG_ROWS_LOG2 = 9;     // 512 blocks of 64 bits (4 KiB per tile); a pair owns 3 bits inside ONE block
constexpr int SIG_TILES = 8;         // k_build_sig: tiles whose signatures one workgroup transposes together (64-B rows)
constexpr int MAX_LISTS_PROF = 4096;
constexpr int SCAN_CHUNK = 256;      // k_scan_skip: consecutive tiles examined by one workgroup at a time (x kt)
constexpr int SCAN_KT_MAX = 4;       // ... kt <= 4 signature tests per thread, so that the whole grid is resident at once
constexpr int scan_kt_max(int nw) { return nw >= 16 ? 2 : 4; } // (a workgroup's hit list lives in LDS: 64 * nw * kt entries)

enum : uint32_t { HALT_NONE = 0, HALT_TABLE_FULL = 1, HALT_POOL_FULL = 2, HALT_VOCAB_FULL = 3, HALT_DELTA_FULL = 4, HALT_RESCAN = 5, HALT_COMM = 6 };

constexpr int KMAX = 16; // merges one sparse launch applies at most (a batch, see select_batch)
struct BatchMerge {
    uint32_t a, b, c, is_new; // the pair, the merged token, whether the merge created it
};
struct DevState {
    uint32_t iter;       // merges recorded so far
    uint32_t done;       // stop rule hit
    uint32_t halt;       // committed: kernels do nothing until the host has serviced it
    uint32_t halt_req;   // raised inside a kernel; committed by the next k_select
    uint32_t a, b, c;    // the merge being applied
    uint32_t c_is_new;
    uint32_t n_tokens;
    uint32_t pool_used;
    uint32_t num_merges; // iteration limit (trainer.py:238)
    uint32_t cand_n;     // length of the candidate list when the last selection ran
    unsigned long long min_freq;
    unsigned long long table_entries;
    unsigned long long live_slots; // sum of tile_len
    unsigned long long sites;      // sites merged since the last k_select
    unsigned long long tokens_now; // T_i
    unsigned long long delta_entries;
    unsigned long long best_count; // count of the merge being applied
    uint32_t xmax;                 // multi-GPU: the largest record count any rank has sent since the host cleared this (k_delta_apply)
    uint32_t kmax;                 // merges the next selection may put in one batch (1: the streaming form applies one merge per launch)
    uint32_t win_shift;            // the selection's window takes the pairs with count >= max - (max >> win_shift); it adapts the shift to
                                   // what it finds (few pairs: wider, more than fit: narrower) -- from counts and distinct pairs only, so
                                   // every rank of a multi-GPU job keeps the same value
    unsigned long long batch_others; // sum of the counts of batch[0 .. n_batch - 1): the sites those merges have in the flat layout
    uint32_t n_batch;              // merges selected and not applied yet: batch[0 .. n_batch), in selection order; a, b, c above = batch[0]
    uint32_t n_select;             // selections that committed a batch so far (= apply launches that had work to do, plus the pending one)
    BatchMerge batch[KMAX];
};

// multi-GPU exchange records: what a rank's apply pass sends to the others (see k_delta_apply)
struct DeltaHdr {
    unsigned long long count; // records this rank produced (may exceed the buffer capacity: overflow, every rank sees it)
    unsigned long long halt;  // this rank's committed halt, if any
};
struct DeltaRec {
    uint32_t key, pad;
    long long delta;
};

// candidate argmax state (see k_argmax_cand / fused_select_tail)
constexpr uint32_t CAND_CAP = 1u << 16; // capacity of the candidate list
struct CandState {
    unsigned long long T;
    uint32_t n, overflow;
    uint32_t n_seen; // entries of the list that were complete when the last selection ran: what a standalone k_argmax_cand
    uint32_t pad;    // may read (n itself moves while kernels that update the table run)
};

struct PairTable {
    uint32_t *keys;
    unsigned long long *cnt;
    uint32_t cap;       // any size >= 2 (not only powers of two: the argmax scan reads every slot, so the table is kept small)
    uint32_t max_probe;
    unsigned long long *entries; // where successful inserts are counted
    // candidate argmax (main table only, else NULL): a count that rises to >= cand_T puts its slot on the candidate list
    CandState *cand_cs;          // list length / overflow flag
    unsigned long long *cand_list; // entries: slot | key << 32
    unsigned long long cand_T;
    // multi-GPU: not a table at all but this rank's send buffer -- every update becomes a (key, delta) record that all
    // ranks add to their replicas after the exchange (k_delta_apply).  keys / cnt are unused then.
    DeltaRec *sink_rec;
    DeltaHdr *sink_hdr;
    uint32_t sink_cap;
};
__device__ __forceinline__ uint32_t hash32(uint32_t k);
__device__ __forceinline__ uint32_t pt_home(const PairTable &t, uint32_t key) { // fast range reduction of the hash
    return (uint32_t)(((unsigned long long)hash32(key) * t.cap) >> 32);
}
__device__ __forceinline__ uint32_t pt_next(const PairTable &t, uint32_t s) { return s + 1 == t.cap ? 0u : s + 1; }

struct Best {
    unsigned long long cnt;
    uint32_t rk;   // lexrank[left] << 16 | lexrank[right]
    uint32_t key;  // left << 16 | right
    uint32_t slot; // table slot of the entry
    uint32_t pad;
};

// A Best record handed from one workgroup to another INSIDE a kernel (k_argmax_cand's last workgroup selects): stored
// and loaded with device-scope accesses, which are coherent across the XCDs' L2s without a cache write-back.
__device__ __forceinline__ void best_store_coherent(Best *p, const Best &b) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
    __hip_atomic_store(q + 0, b.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, ((unsigned long long)b.key << 32) | b.rk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 2, (unsigned long long)b.slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ Best best_load_coherent(const Best *p) {
    unsigned long long *q = reinterpret_cast<unsigned long long *>(const_cast<Best *>(p));
    const unsigned long long a = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long c = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return Best{a, (uint32_t)b, (uint32_t)(b >> 32), (uint32_t)c, 0u};
}

__device__ __forceinline__ uint32_t hash32(uint32_t k) {
    k ^= k >> 16;
    k *= 0x7feb352dU;
    k ^= k >> 15;
    k *= 0x846ca68bU;
    k ^= k >> 16;
    return k;
}

// Tile signatures (skip index).  sig[row * stride + tile], row < SIG_ROWS, one u64 per (row, tile): transposed, so that
// the threads of a workgroup that test consecutive tiles read consecutive words.  A pair hashes to one row and to 3 bit
// positions inside that 64-bit block ("blocked" Bloom filter): the test reads ONE word per tile, and a rewrite that
// creates a pair sets its bits with ONE atomic -- with three separate partitions it took three, on three different cache
// lines, and those scattered read-modify-writes were 9 % of the whole run.  A signature is a SUPERSET of what the tile
// holds: bits are added when a rewrite creates a pair and only a rebuild (k_build_sig) clears stale ones.  With ~1,000
// pairs per tile a block holds ~2 pairs (~9 % of its bits); a tile without the pair passes with probability ~0.15 %.
struct SigHash {
    uint32_t row;
    unsigned long long mask;
};
__device__ __forceinline__ uint32_t fmix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x85EBCA6Bu;
    x ^= x >> 13;
    x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ SigHash sig_hash(uint32_t key) { // key = left << 16 | right
    const uint32_t x = fmix32(key * 0x9E3779B1u + 0x7F4A7C15u);
    SigHash H;
    H.row = x >> (32 - SIG_ROWS_LOG2);
    H.mask = (1ull << (x & 63u)) | (1ull << ((x >> 6) & 63u)) | (1ull << ((x >> 12) & 63u));
    return H;
}
__device__ __forceinline__ void sig_set_pair(unsigned long long *sig, uint32_t stride, uint32_t tile, uint32_t key) {
    const SigHash H = sig_hash(key);
    atomicOr(&sig[(size_t)H.row * stride + tile], H.mask);
}
// The word a scan tests for one tile, as a row pointer (the pair is fixed for the whole launch).
struct SigProbe {
    const unsigned long long *p;
    unsigned long long mask;
    __device__ __forceinline__ bool maybe(uint32_t tile) const { return (p[tile] & mask) == mask; }
};
__device__ __forceinline__ SigProbe sig_probe(const unsigned long long *sig, uint32_t stride, uint32_t key) {
    const SigHash H = sig_hash(key);
    return SigProbe{sig + (size_t)H.row * stride, H.mask};
}

// End of a RARE branch that issued returning global atomics (an update that bypasses the LDS aggregator): wait for them
// here.  Their destination registers are otherwise still "pending" where the branch rejoins the common path, and the
// compiler then guards the next reuse of those registers with s_waitcnt vmcnt(0) ON THE COMMON PATH -- which waits for
// every store and prefetch the wave has in flight (measured: ~1 us per rewritten tile).  (The builtin, not inline asm:
// the wait-count pass has to see it.)  gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15.
__device__ __forceinline__ void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// Within one wave LDS operations complete in order; this keeps the compiler from moving a lane's LDS reads
// above other lanes' LDS writes.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
// how many bits of the wave mask m lie below this lane (v_mbcnt: no per-lane 64-bit mask to keep in registers)
__device__ __forceinline__ uint32_t bits_below_lane(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// inclusive prefix sum over the 64 lanes with DPP moves only (no LDS crossbar): Hillis-Steele inside each row of 16
// lanes (row_shr 1,2,4,8; lanes without a source add 0), then the row totals travel with row_bcast:15 / row_bcast:31.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true); // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true); // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true); // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true); // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
    return x;
}

// ---------------------------------------------------------------- global pair table
// Values another workgroup of the SAME launch may read (the workgroup that finishes last runs the selection, see
// fused_select_tail): device-scope stores and loads, coherent across the XCDs' L2s without a cache write-back.
__device__ __forceinline__ void st_coherent(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_coherent(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_coherent(const unsigned long long *p) { return __hip_atomic_load(const_cast<unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ld_coherent(const uint32_t *p) { return __hip_atomic_load(const_cast<uint32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// The count of slot s rose from `old` to `now`: the add that takes it across the threshold of the candidate argmax puts the slot
// on the list.  Adds to one address are totally ordered, so exactly one adder sees the crossing; a count that falls below the
// threshold and crosses again is listed twice, which is harmless (the same slot, the same count).  (Until round 2 every adder
// that saw a count >= T tried to list it and a bitmap kept the list free of repeats: one more dependent atomic in the flush of
// exactly the workgroups that finish last.)  Invariant kept between two rebuilds of the list: every slot whose count is
// >= cand_T is on it.  Counts only go up through the two functions below, so nothing else has to look at the table.
// (Signed comparisons: a count can be transiently negative while the deltas of one merge arrive in any order.)
__device__ __forceinline__ void cand_note(const PairTable &t, uint32_t s, uint32_t key, unsigned long long old, unsigned long long now) {
    if ((long long)now < (long long)t.cand_T || (long long)old >= (long long)t.cand_T || (long long)now <= 0) return;
    const uint32_t idx = atomicAdd(&t.cand_cs->n, 1u);
    if (idx < CAND_CAP)
        st_coherent(&t.cand_list[idx], (unsigned long long)s | ((unsigned long long)key << 32));
    else
        st_coherent(&t.cand_cs->overflow, 1u);
}
// the count of the slot an entry of the candidate list names
__device__ __forceinline__ unsigned long long *pt_count_ptr(const PairTable &t, unsigned long long entry) { return &t.cnt[(uint32_t)entry]; }
// cnt[s] += d
__device__ __forceinline__ void gt_bump(const PairTable &t, uint32_t s, uint32_t key, long long d) {
    if (d > 0 && t.cand_list) {
        const unsigned long long old = atomicAdd(&t.cnt[s], (unsigned long long)d);
        cand_note(t, s, key, old, old + (unsigned long long)d);
    } else {
        atomicAdd(&t.cnt[s], (unsigned long long)d);
    }
}

// `inserted`: optional per-thread counter of new keys; the caller then adds its wave's total to *t.entries itself (one
// atomic per wave on that one hot address instead of one per new key).
__device__ __forceinline__ void sink_append(const PairTable &t, uint32_t key, long long d) {
    const unsigned long long idx = atomicAdd(&t.sink_hdr->count, 1ull);
    if (idx < t.sink_cap) t.sink_rec[idx] = DeltaRec{key, 0u, d}; // (past the capacity: the count itself tells every rank)
}
__device__ __forceinline__ void gt_add_from(const PairTable &t, DevState *st, uint32_t key, long long d, uint32_t s, uint32_t *inserted) {
    if (t.sink_rec) {
        sink_append(t, key, d);
        return;
    }
    for (uint32_t probe = 0; probe < t.max_probe; ++probe) {
        uint32_t k = __hip_atomic_load(&t.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == EMPTY) {
            k = atomicCAS(&t.keys[s], EMPTY, key);
            if (k == EMPTY) {
                if (inserted) ++*inserted; else atomicAdd(t.entries, 1ull);
                k = key;
            }
        }
        if (k == key) {
            gt_bump(t, s, key, d);
            return;
        }
        s = pt_next(t, s);
    }
    atomicMax(&st->halt_req, (uint32_t)(t.entries == &st->delta_entries ? HALT_DELTA_FULL : HALT_TABLE_FULL));
}
__device__ __forceinline__ void gt_add(const PairTable &t, DevState *st, uint32_t key, long long d, uint32_t *inserted = nullptr) {
    gt_add_from(t, st, key, d, t.sink_rec ? 0u : pt_home(t, key), inserted);
}

#ifdef YB_PROFILE_SCAN
// (diagnostic build) per workgroup of the last launch: [0] everything issued before the flush acknowledged, [1] table keys
// arrived, [2] count updates returned (100 MHz wall clock), [3] entries flushed
constexpr int FLUSH_PROF_BLOCKS = 4096;
__device__ unsigned long long g_flush_prof[FLUSH_PROF_BLOCKS * 4];
#define YB_FLUSH_WAIT_STAMP(i)                                                                  \
    do {                                                                                        \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
        __syncthreads();                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < FLUSH_PROF_BLOCKS) g_flush_prof[blockIdx.x * 4 + (i)] = wall_clock64(); \
    } while (0)
#else
#define YB_FLUSH_WAIT_STAMP(i) do { } while (0)
#endif
// ---------------------------------------------------------------- LDS aggregator (per workgroup)
// V = int for the flat layout (a workgroup's deltas fit 32 bits), unsigned long long for weighted words.
template <class V>
struct Agg {
    uint32_t *keys;
    V *vals;
    uint32_t mask; // entries in use - 1 (AGG_N - 1, or fewer for a merge with few sites: less to initialise and to flush)
};

template <class V>
__device__ __forceinline__ void agg_init(Agg<V> g, int nt = BLOCK) { // nt: threads of the workgroup
    for (int i = threadIdx.x; i <= (int)g.mask; i += nt) {
        g.keys[i] = EMPTY;
        g.vals[i] = (V)0;
    }
}

template <class V>
__device__ __forceinline__ void agg_add(Agg<V> g, const PairTable &t, DevState *st, uint32_t key, long long d) {
    uint32_t s = hash32(key) & g.mask;
#pragma unroll 1
    for (int probe = 0; probe < 8; ++probe) {
        uint32_t k = __hip_atomic_load(&g.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == EMPTY) {
            k = atomicCAS(&g.keys[s], EMPTY, key);
            if (k == EMPTY) k = key;
        }
        if (k == key) {
            atomicAdd(&g.vals[s], (V)d);
            return;
        }
        s = (s + 1) & g.mask;
    }
    gt_add(t, st, key, d); // aggregator full around this hash: go straight to HBM
    vm_drain();            // (rare path; see vm_drain)
}

// slot of `key` in the LDS aggregator (inserting it), or AGG_N when the probe window is full
template <class V>
__device__ __forceinline__ uint32_t agg_slot(Agg<V> g, uint32_t key) {
    uint32_t s = hash32(key) & g.mask;
#pragma unroll 1
    for (int probe = 0; probe < 8; ++probe) {
        uint32_t k = __hip_atomic_load(&g.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == EMPTY) {
            k = atomicCAS(&g.keys[s], EMPTY, key);
            if (k == EMPTY) k = key;
        }
        if (k == key) return s;
        s = (s + 1) & g.mask;
    }
    return AGG_N;
}

// Wave-uniform memo of the last key seen in each of the four delta roles (old/new pair on the left/right of a site)
// and its aggregator slot.  Within one merge nearly every site carries the same four keys, so after the first tile a
// delta costs one fire-and-forget LDS add instead of a probe (load, compare-and-swap, add: three dependent LDS trips).
struct KeyMemo {
    uint32_t key[4];
    uint32_t slot[4];
};

// Flat layout: lanes of one wave usually carry the same few keys (Zipf) -- peel the hottest keys off with
// ballots so that one lane adds the wave's total instead of 64 lanes serialising on one LDS address.
__device__ __forceinline__ void agg_add_wave(Agg<int> g, const PairTable &t, DevState *st, bool valid, uint32_t key,
                                             int sign, int lane, KeyMemo &memo, int role) {
    unsigned long long pend = __ballot(valid);
#pragma unroll 1
    for (int r = 0; r < 3 && pend; ++r) {
        const int leader = __ffsll((long long)pend) - 1;
        const uint32_t k = __builtin_amdgcn_readlane(key, leader);
        const bool mine = valid && key == k;
        const unsigned long long same = __ballot(mine);
        const unsigned long long plus = __ballot(mine && sign > 0);
        const int sum = __popcll(plus) - __popcll(same & ~plus);
        if (sum != 0) {
            uint32_t slot;
            if (memo.key[role] == k) { // wave-uniform
                slot = memo.slot[role];
            } else {
                uint32_t sl = 0;
                if (lane == leader) sl = agg_slot(g, k);
                slot = __builtin_amdgcn_readlane(sl, leader);
                memo.key[role] = k;
                memo.slot[role] = slot;
            }
            if (lane == leader) {
                if (slot < (uint32_t)AGG_N) {
                    atomicAdd(&g.vals[slot], sum);
                } else {
                    gt_add(t, st, k, (long long)sum); // aggregator full around this hash: straight to HBM
                    vm_drain();
                }
            }
        }
        pend &= ~same;
        if (mine) valid = false;
    }
    if (valid) agg_add(g, t, st, key, (long long)sign);
}

// LDS scratch discipline (audited after the race of e85c033).  A __device__ helper that declares its own __shared__ scratch is
// either called once per workgroup and kernel (rank_update_block, fold_block_stats, last_workgroup, select_body, select_eval: their
// first use of the scratch sits behind a barrier of their own) or may be called back to back -- flush_entries' record sink, once per
// role from hist_flush -- and then ENDS on a barrier, so that no wave can rewrite the scratch while a slower one still reads it.
// Helpers that work on the caller's LDS (agg_add / agg_flush, apply_epilogue, slow_tile on a wave's WaveLds) rely on the caller:
// a __syncthreads() between the last add and the flush (apply_epilogue has one), wave_sync() between a wave's own LDS phases.
// tests/test_gpu_distributed.py::test_four_ranks_direct_store_flushes_into_the_send_buffer is the case that exposed the race.
//
// get(i) -> FlushEnt: entry i of the workgroup's delta store (key EMPTY or val 0: nothing); N_ENT entries, NT threads.
struct FlushEnt {
    uint32_t key;
    long long val;
};
template <int NT, int N_ENT, class GetF>
__device__ __forceinline__ void flush_entries(GetF get, const PairTable &t, DevState *st) {
    // every thread owns AGG_N / NT entries; the table keys at their home slots are requested together, so that the
    // usual case (the key sits at its home slot) costs one round trip for all of them.
    constexpr int PER = (N_ENT + NT - 1) / NT;
    uint32_t k[PER], home[PER], tk[PER];
    long long v[PER];
    uint32_t ins = 0;
    if (t.sink_rec) { // multi-GPU: the workgroup's deltas leave as records, one reservation in the send buffer per workgroup
        __shared__ uint32_t s_wtot[NT / 64];
        __shared__ unsigned long long s_sbase;
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = threadIdx.x + q * NT;
            k[q] = EMPTY;
            v[q] = 0;
            if (i < N_ENT) {
                const FlushEnt e = get(i);
                k[q] = e.key;
                v[q] = e.val;
            }
            if (k[q] == EMPTY) v[q] = 0;
            mine += v[q] != 0;
        }
        const uint32_t inc = wave_inclusive_sum(mine);
        const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
        if (lane == 63) s_wtot[wib] = inc;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t tot = 0;
            for (int w = 0; w < NT / 64; ++w) tot += s_wtot[w];
            s_sbase = tot ? atomicAdd(&t.sink_hdr->count, (unsigned long long)tot) : 0ull;
        }
        __syncthreads();
        unsigned long long at = s_sbase + (inc - mine);
        for (int w = 0; w < wib; ++w) at += s_wtot[w];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (v[q] == 0) continue;
            if (at < t.sink_cap) t.sink_rec[at] = DeltaRec{k[q], 0u, v[q]};
            ++at;
        }
        __syncthreads(); // (s_wtot / s_sbase are read above: the direct-indexed store calls this once per role, back to back)
        return;
    }
    YB_FLUSH_WAIT_STAMP(0);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = threadIdx.x + q * NT;
        k[q] = EMPTY;
        v[q] = 0;
        if (i < N_ENT) {
            const FlushEnt e = get(i);
            k[q] = e.key;
            v[q] = e.val;
        }
#ifdef YB_DBG_NOFLUSH
        if (!(k[q] == 12345u && v[q] == 77)) v[q] = 0;
#endif
        if (k[q] == EMPTY) v[q] = 0;
        home[q] = 0;
        tk[q] = EMPTY;
        if (v[q] != 0) {
            home[q] = pt_home(t, k[q]);
            tk[q] = __hip_atomic_load(&t.keys[home[q]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    YB_FLUSH_WAIT_STAMP(1);
#ifdef YB_PROFILE_SCAN
    {
        uint32_t nz = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) nz += v[q] != 0;
        if (blockIdx.x < FLUSH_PROF_BLOCKS) {
            if (threadIdx.x == 0) g_flush_prof[blockIdx.x * 4 + 3] = 0;
            __syncthreads();
            if (nz) atomicAdd(&g_flush_prof[blockIdx.x * 4 + 3], (unsigned long long)nz);
        }
    }
#endif
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (v[q] == 0) continue;
        if (tk[q] == k[q]) {
            gt_bump(t, home[q], k[q], v[q]);
        } else if (tk[q] == EMPTY) {
            gt_add_from(t, st, k[q], v[q], home[q], &ins);
        } else {
            gt_add_from(t, st, k[q], v[q], pt_next(t, home[q]), &ins); // (home holds another key)
        }
    }
    YB_FLUSH_WAIT_STAMP(2);
    // new keys of this wave (ins <= 2 * PER per thread): a few ballots give the total
    uint32_t total = 0;
#pragma unroll
    for (int b = 0; b < 5; ++b) total += (uint32_t)__popcll(__ballot((ins >> b) & 1u)) << b;
    if (total && (threadIdx.x & 63) == 0) atomicAdd(t.entries, (unsigned long long)total);
}

template <class V, int NT = BLOCK> // NT: threads of the workgroup
__device__ __forceinline__ void agg_flush(Agg<V> g, const PairTable &t, DevState *st) {
    flush_entries<NT, AGG_N>(
        [&](int i) -> FlushEnt {
            if ((uint32_t)i > g.mask) return FlushEnt{EMPTY, 0ll};
            return FlushEnt{g.keys[i], (long long)g.vals[i]};
        },
        t, st);
}

// Direct-indexed form of the per-workgroup delta store, for the first merges of a job (few tokens, many sites per tile):
// every delta of a merge (a,b) -> c has one FIXED half -- (x,a) -1, (x,c) +1, (b,y) -1, (c,y) +1 (tile_logic.h: x is the
// left neighbour, or b / c when the neighbour is itself a site; y the right neighbour) -- so the other half indexes one of
// four small LDS arrays directly: a delta is ONE fire-and-forget LDS add, no hashing, no probing, no peeling of hot keys.
constexpr int HIST_V = 512; // tokens that may exist while this form is used (the host checks)
struct Hist {
    int *h; // [4][HIST_V]: role 0 (x,a), 1 (x,c), 2 (b,y), 3 (c,y)
};
__device__ __forceinline__ void hist_init(Hist H) {
    for (int i = threadIdx.x; i < 4 * HIST_V; i += BLOCK) H.h[i] = 0;
}
__device__ __forceinline__ void hist_flush(Hist H, uint32_t a, uint32_t b, uint32_t c, const PairTable &t, DevState *st) {
    // one role at a time (two entries per thread): eight entries per thread in one go do not stay in registers
#pragma unroll 1
    for (uint32_t role = 0; role < 4; ++role) {
        const uint32_t fixed = role == 0 ? a : role == 2 ? b : c;
        flush_entries<BLOCK, HIST_V>(
            [&](int i) -> FlushEnt {
                const int v = H.h[role * HIST_V + i];
                const uint32_t key = role < 2 ? yb_pairkey((uint32_t)i, fixed) : yb_pairkey(fixed, (uint32_t)i);
                return FlushEnt{v ? key : EMPTY, (long long)v};
            },
            t, st);
    }
}

// ---------------------------------------------------------------- tile access helpers
struct TileRegs {
    uint4 va, vb; // slots [8*lane, 8*lane+8) and [512 + 8*lane, ...)
};

__device__ __forceinline__ TileRegs load_tile(const uint16_t *tiles, uint32_t tile, uint32_t len, int lane) {
    const uint4 *base = reinterpret_cast<const uint4 *>(tiles + (size_t)tile * CAP);
    TileRegs r;
    r.va = make_uint4(PADPAD, PADPAD, PADPAD, PADPAD);
    r.vb = r.va;
    if ((uint32_t)(lane * 8) < len) r.va = base[lane];
    if ((uint32_t)(512 + lane * 8) < len) r.vb = base[64 + lane];
    return r;
}

__device__ __forceinline__ bool match4(uint4 v, uint32_t nxt, uint32_t mk) {
    int m = (v.x == mk) | (v.y == mk) | (v.z == mk) | (v.w == mk);
    m |= (int)(__builtin_amdgcn_alignbit(v.y, v.x, 16) == mk);
    m |= (int)(__builtin_amdgcn_alignbit(v.z, v.y, 16) == mk);
    m |= (int)(__builtin_amdgcn_alignbit(v.w, v.z, 16) == mk);
    m |= (int)(__builtin_amdgcn_alignbit(nxt, v.w, 16) == mk);
    return m != 0;
}

// stage: [8 PAD][CAP slots][8 PAD]; position q lives at stage[8 + q]
__device__ __forceinline__ void stage_tile(uint16_t *stg, const TileRegs &r, int lane) {
    *reinterpret_cast<uint4 *>(stg + 8 + lane * 8) = r.va;
    *reinterpret_cast<uint4 *>(stg + 8 + 512 + lane * 8) = r.vb;
}

struct TokAt {
    const uint16_t *stg;
    __device__ __forceinline__ uint32_t operator()(int q) const { return stg[8 + q]; }
};
struct MrgAt {
    const unsigned long long *mb; // mb[0] = 0, mb[1 + k] = sites of round k, mb[1 + rounds] = 0
    __device__ __forceinline__ int operator()(int q) const {
        return (int)((mb[(q + 64) >> 6] >> (q & 63)) & 1ull);
    }
};

// merge-site bitmasks of a staged tile (greedy left-to-right rule, trainer.py:276-285)
__device__ __forceinline__ unsigned long long mark_sites(const uint16_t *stg, unsigned long long *mb, int rounds,
                                                         uint32_t a, uint32_t b, int lane) {
    TokAt T{stg};
    unsigned long long any = 0;
    if (lane == 0) mb[0] = 0ull;
    if (a != b) {
        for (int k = 0; k < rounds; ++k) {
            int p = k * 64 + lane;
            unsigned long long m = __ballot((T(p) == a) && (T(p + 1) == b));
            if (lane == 0) mb[1 + k] = m;
            any |= m;
        }
    } else {
        // runs of a: within a maximal run starting at s, positions s, s+2, ... merge
        int last_non = -1; // wave-uniform: last position < current round whose token is not a
        for (int k = 0; k < rounds; ++k) {
            int p = k * 64 + lane;
            bool isa = T(p) == a;
            unsigned long long non = __ballot(!isa);
            unsigned long long low = non & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
            int q = low ? (k * 64 + 63 - __clzll((long long)low)) : last_non;
            bool mm = isa & (T(p + 1) == a) & (((p - q - 1) & 1) == 0);
            unsigned long long m = __ballot(mm);
            if (lane == 0) mb[1 + k] = m;
            any |= m;
            if (non) last_non = k * 64 + 63 - __clzll((long long)non);
        }
    }
    if (lane == 0) mb[1 + rounds] = 0ull;
    return any;
}

// ================================================================ k_count: pair histogram of the whole stream
// (initial count, trainer.py:227-235; also table rebuilds and the debug recount)
struct CountParams {
    const uint16_t *tiles;
    const uint32_t *tile_len;
    const uint32_t *tile_wbase;
    const uint32_t *wfreq;
    uint32_t n_tiles;
    PairTable table;
    DevState *st;
};

template <bool WEIGHTED>
__global__ __launch_bounds__(BLOCK) void k_count(CountParams P) {
    __shared__ uint32_t s_keys[AGG_N];
    __shared__ unsigned long long s_vals[AGG_N];
    __shared__ __attribute__((aligned(16))) uint16_t s_stage[WPB][8 + CAP + 8];
    Agg<unsigned long long> agg{s_keys, s_vals, (uint32_t)AGG_N - 1u};
    agg_init(agg);
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint16_t *stg = s_stage[wib];
    if (lane < 8) {
        stg[lane] = YB_PAD;
        stg[8 + CAP + lane] = YB_PAD;
    }
    __syncthreads();
    TokAt T{stg};
    const uint32_t stride = gridDim.x * WPB;
    for (uint32_t tile = blockIdx.x * WPB + wib; tile < P.n_tiles; tile += stride) {
        uint32_t len = P.tile_len[tile];
        if (len == 0) continue;
        TileRegs r = load_tile(P.tiles, tile, len, lane);
        wave_sync();
        stage_tile(stg, r, lane);
        wave_sync();
        const int rounds = (len + 63) >> 6;
        uint32_t sep_carry = WEIGHTED ? P.tile_wbase[tile] : 0;
        for (int k = 0; k < rounds; ++k) {
            int p = k * 64 + lane;
            uint32_t x = T(p), y = T(p + 1);
            long long w = 1;
            if (WEIGHTED) {
                unsigned long long sm = __ballot(x == YB_SEP);
                uint32_t widx = sep_carry + bits_below_lane(sm);
                sep_carry += __popcll(sm);
                if (x < YB_PAD && y < YB_PAD) w = (long long)P.wfreq[widx];
            }
            if (x < YB_PAD && y < YB_PAD) agg_add(agg, P.table, P.st, yb_pairkey(x, y), w);
        }
    }
    __syncthreads();
    agg_flush(agg, P.table, P.st);
}

// ================================================================ k_apply: the per-merge pass
struct ApplyParams {
    uint16_t *tiles;
    uint32_t *tile_len;
    const uint32_t *tile_wbase;
    const uint32_t *wfreq;
    uint32_t n_tiles;
    PairTable out; // where deltas go: the pair table (1 GPU) or this rank's send buffer as records (multi-GPU: PairTable::sink_*)
    DevState *st;
    unsigned long long *blk_stats; // [2 * gridDim.x]: sites merged, slots freed per workgroup (plain stores)
    unsigned long long *sig;       // tile signatures (may be NULL)
    uint32_t sig_stride;
    uint32_t agg_mask;             // k_scan_skip: LDS aggregator entries in use - 1 (a power of two - 1, <= AGG_N - 1)
    uint32_t stats_fresh;          // 1: this launch is the only one of its merge that writes blk_stats (the selection cleared them):
                                   // a workgroup stores its counters without reading the slot first (a dependent load in front of every flush)
};

// lane i <- lane i+1's value, lane 63 <- fill (one DPP move, no LDS crossbar)
__device__ __forceinline__ uint32_t next_lane(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}

// bit j = "the pair starting at element j of this 8-slot segment equals (a,b)"
__device__ __forceinline__ uint32_t match_mask8(uint4 v, uint32_t nxt, uint32_t mk) {
    uint32_t m = (uint32_t)(v.x == mk);
    m |= (uint32_t)(__builtin_amdgcn_alignbit(v.y, v.x, 16) == mk) << 1;
    m |= (uint32_t)(v.y == mk) << 2;
    m |= (uint32_t)(__builtin_amdgcn_alignbit(v.z, v.y, 16) == mk) << 3;
    m |= (uint32_t)(v.z == mk) << 4;
    m |= (uint32_t)(__builtin_amdgcn_alignbit(v.w, v.z, 16) == mk) << 5;
    m |= (uint32_t)(v.w == mk) << 6;
    m |= (uint32_t)(__builtin_amdgcn_alignbit(nxt, v.w, 16) == mk) << 7;
    return m;
}

// bit j = "element j of this 8-slot segment equals the 16-bit value x"
__device__ __forceinline__ uint32_t eq_mask8(uint4 v, uint32_t x) {
    uint32_t m = (uint32_t)((v.x & 0xffffu) == x) | (uint32_t)((v.x >> 16) == x) << 1;
    m |= (uint32_t)((v.y & 0xffffu) == x) << 2 | (uint32_t)((v.y >> 16) == x) << 3;
    m |= (uint32_t)((v.z & 0xffffu) == x) << 4 | (uint32_t)((v.z >> 16) == x) << 5;
    m |= (uint32_t)((v.w & 0xffffu) == x) << 6 | (uint32_t)((v.w >> 16) == x) << 7;
    return m;
}

__device__ __forceinline__ uint32_t elem16(const uint4 &v, int j) {
    const uint32_t d = (j >> 1) == 0 ? v.x : (j >> 1) == 1 ? v.y : (j >> 1) == 2 ? v.z : v.w;
    return (j & 1) ? (d >> 16) : (d & 0xffffu);
}

// per-wave LDS scratch of the slow path
struct WaveLds {
    uint16_t stage[8 + CAP + 8];       // tokens with PAD halo; position q at stage[8 + q].  The compacted tile is
                                       // built in place at stage + 8 once the last neighbour lookup is done.
    unsigned long long mb[CAP / 64 + 2]; // a == b only: site bitmap by 64-position rounds
    uint32_t mbits[1 + CAP / 32 + 1];  // site bitmap, word w at mbits[1 + w], zero halo words
    uint32_t dmask[CAP / 32 + 1];      // drop bitmap
    uint32_t dpref[CAP / 32 + 1];      // exclusive popcount prefix of dmask
    uint32_t smask[CAP / 32];          // weighted: SEP bitmap
    uint32_t spref[CAP / 32];          // weighted: exclusive popcount prefix of smask
#ifdef YB_PROFILE_SLOW
    unsigned long long prof[8];
#endif
};

struct MrgBits {
    const uint32_t *mbits;
    __device__ __forceinline__ int operator()(int q) const { return (int)((mbits[1 + (q >> 5)] >> (q & 31)) & 1u); }
};

// exclusive prefix of popcounts over the 32 words of a bitmap (lanes 0..31 hold one word each)
__device__ __forceinline__ uint32_t bitmap_prefix(uint32_t word, int lane, uint32_t *total) {
    const uint32_t pc = lane < 32 ? __popc(word) : 0u;
    const uint32_t inc = wave_inclusive_sum(pc);
    *total = __builtin_amdgcn_readlane(inc, 63);
    return inc - pc;
}

// ================================================================ k_count_bytes: initial count of the flat layout
// At load time every token is a byte, so the key space is exactly 256 x 256.  Four passes, pass q owning the
// pairs whose left byte is in [64q, 64q+64): a 64 KiB dense u32 histogram per workgroup in LDS (no hashing, no
// HBM atomics in the loop), flushed into a dense u64 array, which k_dense_to_table then inserts into the table.
constexpr int CB_BLOCK = 1024;
constexpr int CB_WAVES = CB_BLOCK / 64;
struct CountBytesParams {
    const uint16_t *tiles;
    const uint32_t *tile_len;
    uint32_t n_tiles;
    unsigned long long *dense; // [65536]
    uint32_t blocks_per_pass;
};

__global__ __launch_bounds__(CB_BLOCK) void k_count_bytes(CountBytesParams P) {
    __shared__ uint32_t s_hist[64 * 256];
    const uint32_t pass = blockIdx.x / P.blocks_per_pass;
    const uint32_t bip = blockIdx.x % P.blocks_per_pass;
    for (int i = threadIdx.x; i < 64 * 256; i += CB_BLOCK) s_hist[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t stride = P.blocks_per_pass * CB_WAVES;
    for (uint32_t tile = bip * CB_WAVES + wib; tile < P.n_tiles; tile += stride) {
        const uint32_t len = P.tile_len[tile];
        if (len == 0) continue;
        const TileRegs r = load_tile(P.tiles, tile, len, lane);
        const uint32_t b0 = __builtin_amdgcn_readfirstlane(r.vb.x);
        const uint32_t na = next_lane(r.va.x, b0);
        const uint32_t nb = next_lane(r.vb.x, PADPAD);
#pragma unroll
        for (int seg = 0; seg < 2; ++seg) {
            const uint4 &v = seg ? r.vb : r.va;
            const uint32_t nx = (seg ? nb : na) & 0xffffu;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t x = elem16(v, j);
                const uint32_t y = j < 7 ? elem16(v, j + 1) : nx;
                if ((x >> 6) == pass && y < 256u) atomicAdd(&s_hist[((x & 63u) << 8) | y], 1u); // x < 256 implied
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 256; i += CB_BLOCK) {
        const uint32_t v = s_hist[i];
        if (v) atomicAdd(&P.dense[pass * 16384u + i], (unsigned long long)v);
    }
}

struct DenseToTableParams {
    const unsigned long long *dense;
    PairTable table;
    DevState *st;
};
__global__ __launch_bounds__(BLOCK) void k_dense_to_table(DenseToTableParams P) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x; // i = left * 256 + right
    if (i >= 65536u) return;
    const unsigned long long v = P.dense[i];
    if (v) gt_add(P.table, P.st, yb_pairkey(i >> 8, i & 255u), (long long)v);
}

#ifdef YB_PROFILE_SLOW
__device__ unsigned long long g_slow_prof[8]; // [0] tiles, [1..5] cycles per phase
#define YB_STAMP(i)                                                                     \
    do {                                                                                \
        unsigned long long t__ = __builtin_readcyclecounter();                          \
        if (lane == 0) W.prof[i] += t__ - t_prev;                                       \
        t_prev = t__;                                                                   \
    } while (0)
#else
#define YB_STAMP(i) do { } while (0)
#endif

// What a workgroup needs to run the slow path
template <class AggV>
struct SlowCtx {
    ApplyParams P;
    Agg<AggV> agg;
    DevState *st;
    uint32_t a, b, c, mk, self;
    int lane;
    KeyMemo memo; // wave-uniform
    int *hist;    // != NULL: the direct-indexed delta store (Hist) is in use instead of agg
};

// slot q of the tile (wave-uniform q), read out of the registers that hold the tile: lane (q>>3)&63 owns it
__device__ __forceinline__ uint32_t tile_elem_uniform(const TileRegs &r, int q) {
    if (q < 0 || q >= CAP) return YB_PAD;
    const bool segB = q >= 512;
    const int l = (q >> 3) & 63, e = q & 7;
    // (values first, selects second: a conditional between the two MEMBERS makes the compiler keep r in scratch)
    const uint32_t ax = r.va.x, ay = r.va.y, az = r.va.z, aw = r.va.w;
    const uint32_t bx = r.vb.x, by = r.vb.y, bz = r.vb.z, bw = r.vb.w;
    const uint32_t x = segB ? bx : ax, y = segB ? by : ay, z = segB ? bz : az, w4 = segB ? bw : aw;
    const uint32_t d01 = (e & 2) ? y : x, d23 = (e & 2) ? w4 : z;
    const uint32_t d = (e & 4) ? d23 : d01;
    const uint32_t w = __builtin_amdgcn_readlane(d, l);
    return (e & 1) ? (w >> 16) : (w & 0xffffu);
}

#ifdef YB_PROFILE_SCAN
__device__ unsigned long long g_ss_prof[8];
#define YB_SS_STAMP(i)                                                          \
    do {                                                                        \
        const unsigned long long t_now = __builtin_readcyclecounter();          \
        if (lane == 0 && blockIdx.x == 7) atomicAdd(&g_ss_prof[i], t_now - t_ss);                  \
        t_ss = __builtin_readcyclecounter();                                    \
    } while (0)
#else
#define YB_SS_STAMP(i) do { } while (0)
#endif
// The common case of the rewrite (flat layout, a != b): the tile holds exactly ONE site.  Everything is wave-uniform:
// the site and its neighbours come out of the registers with v_readlane, the four deltas go through the key memo, and
// the compaction is a funnel shift in registers -- the slots that leave the tile are one contiguous run (the b, or the
// whole word (c, b, SEP) when the word was exactly (a b)), so every later slot moves left by the same s in {1, 3}.
// No LDS staging, no bitmaps, no scatter.
// WEIGHTED (pooled words): the site's word carries a frequency -- the word's index is tile_wbase + the number of SEP slots in
// front of the site, counted in registers; nothing but the b leaves the tile (words are never dropped in this layout).
// DEFER (batched sparse launch: several merges may rewrite the same tile one after the other): nothing is stored -- the
// rewritten tile replaces `r`, `len` becomes its new length and `first_changed` the first slot that differs from HBM; the
// caller writes the tile back once, after the last merge of the batch has seen it.
template <class AggV, bool HIST = false, bool WEIGHTED = false, bool DEFER = false>
__device__ __forceinline__ void single_site_tile(SlowCtx<AggV> &C, uint32_t tile, uint32_t &len, TileRegs &r,
                                                 int lane_s, uint32_t mm_s, unsigned long long &wave_sites,
                                                 unsigned long long &wave_freed, uint32_t &first_changed) {
    const ApplyParams &P = C.P;
    DevState *st = C.st;
    const uint32_t a = C.a, b = C.b, c = C.c;
    const int lane = C.lane;
#ifdef YB_PROFILE_SCAN
    unsigned long long t_ss = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 7) atomicAdd(&g_ss_prof[0], 1ull);
#endif
    const int j = __ffs((int)mm_s) - 1;
    const int p = j < 8 ? lane_s * 8 + j : 512 + lane_s * 8 + (j - 8); // the site (wave-uniform)
    const uint32_t L = tile_elem_uniform(r, p - 1), R = tile_elem_uniform(r, p + 2);
    const bool left = L < YB_PAD, right = R < YB_PAD;
    const bool dead = !WEIGHTED && !left && R == YB_SEP; // flat layout: the word was exactly (a b): it leaves the stream (yb_site_word_dies)
    long long w = 1; // the word's frequency
    if constexpr (WEIGHTED) {
        // SEP slots in front of position p: whole groups of the lanes before the site's, the low slots of the site's own group,
        // all of segment A when the site lies in segment B -- one wave sum, then the frequency is on its way while the tile is rewritten
        const uint32_t sepA = eq_mask8(r.va, YB_SEP), sepB = eq_mask8(r.vb, YB_SEP);
        const bool inB = p >= 512;
        const uint32_t mine = inB ? sepB : sepA, below = (1u << (p & 7)) - 1u;
        uint32_t cnt = lane < lane_s ? __popc(mine) : lane == lane_s ? __popc(mine & below) : 0u;
        if (inB) cnt += __popc(sepA);
        cnt = wave_inclusive_sum(cnt);
        const uint32_t widx = P.tile_wbase[tile] + (uint32_t)__builtin_amdgcn_readlane(cnt, 63);
        w = (long long)P.wfreq[widx];
    }

    YB_SS_STAMP(1);
    // The two pairs this site creates will be present in the tile: one signature word each, set by lanes 0 and 1 -- FIRST, as
    // soon as the neighbours are known: these are the wave's only memory operations here (DEFER form), and the candidate loop
    // cannot take the next tile's prefetched registers before they are acknowledged (the in-order vmcnt leaves the compiler no
    // other choice): the compaction and the LDS deltas below run meanwhile.  (Non-DEFER form: the stores below go out next.)
    if (P.sig && !dead && lane < 2) {
        if (lane ? right : left) sig_set_pair(P.sig, P.sig_stride, tile, lane ? yb_pairkey(c, R) : yb_pairkey(L, c));
    }
    // ---- compaction in registers
    const int D0 = dead ? p : p + 1; // first slot that leaves
    const int s = dead ? 3 : 1;      // how many leave (contiguous)
    const uint32_t new_len = len - (uint32_t)s;
    const uint32_t pad_end = (new_len + 7u) & ~7u;
    uint4 *wb = reinterpret_cast<uint4 *>(P.tiles + (size_t)tile * CAP);
    (void)wb;
    (void)pad_end;
    uint4 ra = r.va, rb = r.vb; // DEFER: the rewritten tile
    // the first dwords of segment B (for segment A's lane 63)
    const uint32_t fx = __builtin_amdgcn_readfirstlane(r.vb.x), fy = __builtin_amdgcn_readfirstlane(r.vb.y);
    const uint32_t fz = __builtin_amdgcn_readfirstlane(r.vb.z);
#pragma unroll
    for (int seg = 0; seg < 2; ++seg) {
        if (seg == 0 && p >= 512) continue; // the site and everything that moves is in segment B (uniform)
        const int g0 = seg ? 512 + lane * 8 : lane * 8;
        const uint4 v = seg ? r.vb : r.va;
        // window of 7 dwords: own 4 + the next group's first 3 (enough for s <= 3)
        const uint32_t n0 = next_lane(v.x, seg ? PADPAD : fx);
        const uint32_t n1 = next_lane(v.y, seg ? PADPAD : fy);
        const uint32_t n2 = next_lane(v.z, seg ? PADPAD : fz);
        uint32_t U0 = v.x, U1 = v.y, U2 = v.z, U3 = v.w;
        { // the lane whose group holds the site: a -> c.  Branch-free: the half-word mask and the value are wave-uniform
          // (p is), only "is it my dword" differs by lane -- as nested ifs this was ~90 instructions of exec-mask juggling
            const int e = p - g0; // (0..7 in the one lane that holds the site)
            const uint32_t hm = (p & 1) ? 0xffff0000u : 0x0000ffffu, cv = (p & 1) ? (c << 16) : c;
            const uint32_t sel = (!dead && e >= 0 && e < 8) ? hm : 0u; // 0: nothing changes in this lane
            const int dw = e >> 1;
            const uint32_t m0 = dw == 0 ? sel : 0u, m1 = dw == 1 ? sel : 0u, m2 = dw == 2 ? sel : 0u, m3 = dw == 3 ? sel : 0u;
            U0 = (U0 & ~m0) | (cv & m0);
            U1 = (U1 & ~m1) | (cv & m1);
            U2 = (U2 & ~m2) | (cv & m2);
            U3 = (U3 & ~m3) | (cv & m3);
        }
        uint32_t S0, S1, S2, S3; // the group as it looks when every slot comes from s positions to the right
        if (s == 1) {
            S0 = __builtin_amdgcn_alignbit(v.y, v.x, 16);
            S1 = __builtin_amdgcn_alignbit(v.z, v.y, 16);
            S2 = __builtin_amdgcn_alignbit(v.w, v.z, 16);
            S3 = __builtin_amdgcn_alignbit(n0, v.w, 16);
        } else {
            S0 = __builtin_amdgcn_alignbit(v.z, v.y, 16);
            S1 = __builtin_amdgcn_alignbit(v.w, v.z, 16);
            S2 = __builtin_amdgcn_alignbit(n0, v.w, 16);
            S3 = __builtin_amdgcn_alignbit(n1, n0, 16);
        }
        (void)n2;
        // slots before D0 keep their place, slots from D0 on take the shifted value
        const int bd = D0 - g0; // number of leading slots of this group that stay
        auto mix = [&](uint32_t U, uint32_t S, int i) -> uint32_t {
            const uint32_t m = ((2 * i < bd) ? 0x0000ffffu : 0u) | ((2 * i + 1 < bd) ? 0xffff0000u : 0u);
            return (U & m) | (S & ~m);
        };
        const uint4 o = make_uint4(mix(U0, S0, 0), mix(U1, S1, 1), mix(U2, S2, 2), mix(U3, S3, 3));
        if constexpr (DEFER) {
            if (seg) rb = o; else ra = o; // (a group in front of the site comes out unchanged: bd >= 8 keeps every slot)
        } else {
            if (g0 + 8 > p && (uint32_t)g0 < pad_end) wb[(seg ? 64 : 0) + lane] = o; // groups from the first changed slot on
        }
    }
    if constexpr (DEFER) {
        r.va = ra;
        r.vb = rb;
        len = new_len;
        first_changed = min(first_changed, (uint32_t)p);
    } else {
        if (lane == 0) P.tile_len[tile] = new_len;
    }
    YB_SS_STAMP(4);
    YB_SS_STAMP(3);
    // ---- deltas (tile_logic.h with no neighbouring site): (L,a)-1 (L,c)+1 (b,R)-1 (c,R)+1, one per lane: the four
    // aggregator probes run side by side instead of one after the other
    {
        const uint32_t kl = lane & 1 ? yb_pairkey(L, c) : yb_pairkey(L, a);
        const uint32_t kr = lane & 1 ? yb_pairkey(c, R) : yb_pairkey(b, R);
        const bool on = lane < 4 && (lane < 2 ? left : right);
        if constexpr (HIST) { // lane = role: (L,a) -1, (L,c) +1, (b,R) -1, (c,R) +1
            if (on) atomicAdd(&C.hist[lane * HIST_V + (int)(lane < 2 ? L : R)], (lane & 1) ? 1 : -1);
        } else {
            if (on) agg_add(C.agg, P.out, st, lane < 2 ? kl : kr, (lane & 1) ? w : -w);
        }
    }
    YB_SS_STAMP(2);
    wave_sites += 1;
    wave_freed += (unsigned long long)s;
}

// Rewrites one tile that contains at least one candidate site: pair-count deltas, drop bitmap, compaction,
// write-back.  Work is proportional to the number of sites.  Returns false when nothing changed.
// HAVE_MASKS: the caller has the candidate masks of the two segments already (match_mask8): pmA, pmB.
// DEFER: see single_site_tile -- the compacted tile is read back from the LDS image into `r` instead of being stored.
template <bool WEIGHTED, class AggV, bool HIST = false, bool HAVE_MASKS = false, bool DEFER = false>
__device__ __forceinline__ bool slow_tile(SlowCtx<AggV> &C, WaveLds &W, uint32_t tile, uint32_t &len_io, TileRegs &r,
                                          uint32_t na, uint32_t nb, unsigned long long &wave_sites,
                                          unsigned long long &wave_freed, uint32_t &first_changed, uint32_t pmA = 0u, uint32_t pmB = 0u) {
    const uint32_t len = len_io;
    const ApplyParams &P = C.P;
    DevState *st = C.st;
    const uint32_t a = C.a, b = C.b, c = C.c, mk = C.mk, self = C.self;
    const int lane = C.lane;
    uint16_t *stg = W.stage;
    TokAt T{stg};
    MrgBits M{W.mbits};
    const int pA = lane * 8, pB = 512 + lane * 8; // first positions of this lane's two segments
#ifdef YB_PROFILE_SLOW
    unsigned long long t_prev = __builtin_readcyclecounter();
    if (lane == 0) W.prof[0] += 1ull;
#endif
    uint32_t mA, mB; // site masks of the two segments
    if (a != b) {
        mA = HAVE_MASKS ? pmA : match_mask8(r.va, na, mk);
        mB = HAVE_MASKS ? pmB : match_mask8(r.vb, nb, mk);
        if (!WEIGHTED) {
            const uint32_t mine = mA | (mB << 8);
            const unsigned long long holders = __ballot(mine != 0);
            if (__popcll(holders) == 1) {
                const int lane_s = __ffsll((long long)holders) - 1;
                const uint32_t mm_s = __builtin_amdgcn_readlane(mine, lane_s);
                if (__popc(mm_s) == 1) {
                    single_site_tile<AggV, HIST, false, DEFER>(C, tile, len_io, r, lane_s, mm_s, wave_sites, wave_freed, first_changed);
                    return true;
                }
            }
        }
        wave_sync();
        stage_tile(stg, r, lane);
    } else {
        wave_sync();
        stage_tile(stg, r, lane); // greedy parity rule over runs of a (trainer.py:276-285): through the round bitmaps
        wave_sync();
        const int rounds = (len + 63) >> 6;
        const unsigned long long any = mark_sites(stg, W.mb, rounds, a, b, lane);
        wave_sync();
        if (!any) return false; // a lone candidate can lose to the parity rule -- never for a != b
        mA = (pA >> 6) < rounds ? (uint32_t)(W.mb[1 + (pA >> 6)] >> (pA & 63)) & 0xffu : 0u;
        mB = (pB >> 6) < rounds ? (uint32_t)(W.mb[1 + (pB >> 6)] >> (pB & 63)) & 0xffu : 0u;
    }
    // site bitmap (for M(p-2), M(p+2) lookups) and drop bitmap seeded with the PAD elements of the live prefix
    reinterpret_cast<uint8_t *>(W.mbits + 1)[lane] = (uint8_t)mA;
    reinterpret_cast<uint8_t *>(W.mbits + 1)[64 + lane] = (uint8_t)mB;
    uint32_t dA = eq_mask8(r.va, YB_PAD), dB = eq_mask8(r.vb, YB_PAD);
    if ((uint32_t)pA + 8 > len) dA &= (uint32_t)pA < len ? (1u << (len - pA)) - 1u : 0u;
    if ((uint32_t)pB + 8 > len) dB &= (uint32_t)pB < len ? (1u << (len - pB)) - 1u : 0u;
    reinterpret_cast<uint8_t *>(W.dmask)[lane] = (uint8_t)dA;
    reinterpret_cast<uint8_t *>(W.dmask)[64 + lane] = (uint8_t)dB;
    if (lane == 0) W.dmask[CAP / 32] = 0;
    uint32_t wbase = 0;
    if (WEIGHTED) {
        reinterpret_cast<uint8_t *>(W.smask)[lane] = (uint8_t)eq_mask8(r.va, YB_SEP);
        reinterpret_cast<uint8_t *>(W.smask)[64 + lane] = (uint8_t)eq_mask8(r.vb, YB_SEP);
        wbase = P.tile_wbase[tile];
    }
    wave_sync();
    if (WEIGHTED) {
        uint32_t tot;
        uint32_t pre = bitmap_prefix(lane < 32 ? W.smask[lane & 31] : 0u, lane, &tot);
        if (lane < 32) W.spref[lane] = pre;
        wave_sync();
    }
    YB_STAMP(1);
    // ---- deltas + drop marks: each lane walks its own sites (usually none or one)
    uint32_t mm = mA | (mB << 8);
    {
        uint32_t tot = __popc(mm);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) tot += __shfl_xor(tot, o);
        wave_sites += tot;
    }
    while (__any(mm != 0)) {
        const bool site = mm != 0;
        const int j = site ? (__ffs((int)mm) - 1) : 0;
        mm &= mm - 1;
        const int p = j < 8 ? pA + j : pB + (j - 8);
        YbDeltas d;
        d.left = d.right = false;
        d.lo = d.ln = d.ro = d.rn = 0;
        if (site) {
            yb_site_deltas(p, a, b, c, T, M, d);
            if (P.sig) { // the pairs this site creates are now present in the tile
                if (d.left) sig_set_pair(P.sig, P.sig_stride, tile, d.ln);
                if (d.right) sig_set_pair(P.sig, P.sig_stride, tile, d.rn);
            }
            atomicOr(&W.dmask[(p + 1) >> 5], 1u << ((p + 1) & 31));
            if (!WEIGHTED && yb_site_word_dies(p, T)) {
                atomicOr(&W.dmask[p >> 5], 1u << (p & 31));
                atomicOr(&W.dmask[(p + 2) >> 5], 1u << ((p + 2) & 31));
            }
        }
        if constexpr (WEIGHTED) {
            if (site) {
                const uint32_t sw = W.smask[p >> 5];
                const uint32_t widx = wbase + W.spref[p >> 5] + __popc(sw & ((1u << (p & 31)) - 1u));
                const long long w = (long long)P.wfreq[widx];
                if (d.left) {
                    if (d.lo != self) agg_add(C.agg, P.out, st, d.lo, -w);
                    agg_add(C.agg, P.out, st, d.ln, +w);
                }
                if (d.right) {
                    if (d.ro != self) agg_add(C.agg, P.out, st, d.ro, -w);
                    agg_add(C.agg, P.out, st, d.rn, +w);
                }
            }
        } else if constexpr (HIST) { // (every delta has one fixed half: the other one is the index -- see Hist)
            static_assert(!WEIGHTED, "the direct-indexed store holds 32-bit deltas");
            if (d.left && d.lo != self) atomicAdd(&C.hist[0 * HIST_V + (int)(d.lo >> 16)], -1);
            if (d.left) atomicAdd(&C.hist[1 * HIST_V + (int)(d.ln >> 16)], 1);
            if (d.right && d.ro != self) atomicAdd(&C.hist[2 * HIST_V + (int)(d.ro & 0xffffu)], -1);
            if (d.right) atomicAdd(&C.hist[3 * HIST_V + (int)(d.rn & 0xffffu)], 1);
        } else {
            agg_add_wave(C.agg, P.out, st, d.left && d.lo != self, d.lo, -1, lane, C.memo, 0);
            agg_add_wave(C.agg, P.out, st, d.left, d.ln, +1, lane, C.memo, 1);
            agg_add_wave(C.agg, P.out, st, d.right && d.ro != self, d.ro, -1, lane, C.memo, 2);
            agg_add_wave(C.agg, P.out, st, d.right, d.rn, +1, lane, C.memo, 3);
        }
    }
    wave_sync();
    YB_STAMP(2);
    // ---- compaction: every kept element moves left by the number of dropped elements before it.  Out of the registers:
    // a lane still holds its 2 x 8 slots; it counts what it keeps, one DPP prefix sum over the lanes gives its two output
    // offsets (segment A's slots come first), and it scatters its kept slots into the LDS image -- 16 independent 2-byte
    // writes per lane and no round trip in between.  (Position-parallel rounds over the staged tile, 64 slots at a time,
    // were 32 dependent LDS trips per tile: the bound of the dense phase.)
    uint16_t *outb = W.stage + 8; // (all reads of the staged tokens are done: the site loop is behind us)
    const uint32_t dA2 = reinterpret_cast<const uint8_t *>(W.dmask)[lane], dB2 = reinterpret_cast<const uint8_t *>(W.dmask)[64 + lane];
    const uint32_t vA = (uint32_t)pA >= len ? 0u : ((uint32_t)pA + 8u <= len ? 0xffu : (1u << (len - (uint32_t)pA)) - 1u);
    const uint32_t vB = (uint32_t)pB >= len ? 0u : ((uint32_t)pB + 8u <= len ? 0xffu : (1u << (len - (uint32_t)pB)) - 1u);
    const uint32_t keepA = ~dA2 & vA, keepB = ~dB2 & vB;
    const uint32_t kA = __popc(keepA), kB = __popc(keepB);
    const uint32_t inc = wave_inclusive_sum(kA | (kB << 16));
    const uint32_t tot = __builtin_amdgcn_readlane(inc, 63);
    const uint32_t new_len = (tot & 0xffffu) + (tot >> 16);
    const uint32_t dropped = len - new_len;
    uint32_t first_drop = CAP; // first changed position: 16-B groups before it are unchanged in HBM
    {
        const uint32_t dw = lane < 32 ? W.dmask[lane & 31] : 0u;
        const unsigned long long nz = __ballot(dw != 0);
        if (nz) {
            const int wl = __ffsll((long long)nz) - 1;
            first_drop = wl * 32 + (__ffs((int)__shfl(dw, wl)) - 1);
        }
        // a site at p rewrites slot p (a -> c) and drops p+1: the first CHANGED slot can be first_drop - 1
        if (first_drop > 0) first_drop -= 1;
    }
    {
        // (this scatter was 144 of the rewrite's ~265 VALU instructions per tile, and the streaming phase is VALU-bound:
        // an element's offset is one and + one popcount-with-add instead of a running sum, the token goes out as it lies
        // in its dword, and the few sites get their merged token in a second, short step)
        const uint32_t oA = (inc & 0xffffu) - kA, oB = (tot & 0xffffu) + (inc >> 16) - kB;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if ((keepA >> j) & 1u) outb[oA + __popc(keepA & ((1u << j) - 1u))] = (uint16_t)elem16(r.va, j);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if ((keepB >> j) & 1u) outb[oB + __popc(keepB & ((1u << j) - 1u))] = (uint16_t)elem16(r.vb, j);
        for (uint32_t sm = mA; __any(sm != 0u); sm &= sm - 1u) { // (a site's a becomes c -- unless its whole word leaves)
            const uint32_t lb = sm & (0u - sm);
            if (lb & keepA) outb[oA + __popc(keepA & (lb - 1u))] = (uint16_t)c;
        }
        for (uint32_t sm = mB; __any(sm != 0u); sm &= sm - 1u) {
            const uint32_t lb = sm & (0u - sm);
            if (lb & keepB) outb[oB + __popc(keepB & (lb - 1u))] = (uint16_t)c;
        }
    }
    const uint32_t pad_end = (new_len + 7u) & ~7u;
    if (lane < 8 && new_len + lane < pad_end) outb[new_len + lane] = YB_PAD;
    wave_sync();
    YB_STAMP(3);
    if constexpr (DEFER) {
        const uint4 pp = make_uint4(PADPAD, PADPAD, PADPAD, PADPAD);
        r.va = (uint32_t)pA < new_len ? *reinterpret_cast<const uint4 *>(outb + pA) : pp;
        r.vb = (uint32_t)pB < new_len ? *reinterpret_cast<const uint4 *>(outb + pB) : pp;
        len_io = new_len;
        first_changed = min(first_changed, first_drop);
        wave_sync(); // (the image is read: the next rewrite of this wave may stage over it)
    } else {
        uint4 *wb = reinterpret_cast<uint4 *>(P.tiles + (size_t)tile * CAP);
        if ((uint32_t)pA < new_len && (uint32_t)pA + 8 > first_drop) wb[lane] = *reinterpret_cast<const uint4 *>(outb + pA);
        if ((uint32_t)pB < new_len && (uint32_t)pB + 8 > first_drop) wb[64 + lane] = *reinterpret_cast<const uint4 *>(outb + pB);
        if (lane == 0) P.tile_len[tile] = new_len;
    }
    wave_freed += dropped;
    YB_STAMP(4);
    return true;
}

__device__ __forceinline__ void wave_lds_init(WaveLds &W, int lane) {
    if (lane < 8) {
        W.stage[lane] = YB_PAD;
        W.stage[8 + CAP + lane] = YB_PAD;
    }
    if (lane == 0) {
        W.mbits[0] = 0;
        W.mbits[1 + CAP / 32] = 0;
    }
#ifdef YB_PROFILE_SLOW
    if (lane < 8) W.prof[lane] = 0ull;
#endif
}

// per-workgroup epilogue shared by k_apply and k_slow: counters by plain stores, deltas to the table
template <class AggV, int NT = BLOCK>
__device__ __forceinline__ void apply_epilogue(const ApplyParams &P, Agg<AggV> agg, DevState *st, unsigned long long *s_cnt,
                                               unsigned long long wave_sites, unsigned long long wave_freed, int lane,
                                               int *hist = nullptr, uint32_t ha = 0, uint32_t hb = 0, uint32_t hc = 0) {
    if (lane == 0) {
        if (wave_sites) atomicAdd(&s_cnt[0], wave_sites);
        if (wave_freed) atomicAdd(&s_cnt[1], wave_freed);
    }
    __syncthreads();
    if (threadIdx.x == 0 && (s_cnt[0] | s_cnt[1])) { // one slot per workgroup, no atomics; the selection sums and clears them
        if (P.stats_fresh) {
            st_coherent(&P.blk_stats[2 * blockIdx.x], s_cnt[0]); // (the selection may run in this very launch)
            st_coherent(&P.blk_stats[2 * blockIdx.x + 1], s_cnt[1]);
        } else {
            st_coherent(&P.blk_stats[2 * blockIdx.x], P.blk_stats[2 * blockIdx.x] + s_cnt[0]);
            st_coherent(&P.blk_stats[2 * blockIdx.x + 1], P.blk_stats[2 * blockIdx.x + 1] + s_cnt[1]);
        }
    }
    if (hist) // (the direct-indexed store of merge (ha, hb) -> hc)
        hist_flush(Hist{hist}, ha, hb, hc, P.out, st);
    else
        agg_flush<AggV, NT>(agg, P.out, st);
}

// ---------------------------------------------------------------- token byte strings on the device
struct TokRec {      // 32 B per token: what the selection needs about a token, in one place
    uint32_t rank;   // lexrank[id]: rank of the token's bytes in Python bytes order among all tokens
    uint32_t len;    // bytes
    unsigned long long hash; // yb_hash_bytes of its bytes
    unsigned long long pre8; // its first eight bytes, big-endian, zero padded: integer order = bytes order as far as they decide
    unsigned long long pad;
};
YB_HD unsigned long long yb_pre8(const uint8_t *p, uint32_t n) {
    unsigned long long v = 0;
    for (uint32_t i = 0; i < 8u; ++i) v = (v << 8) | (i < n ? (unsigned long long)p[i] : 0ull);
    return v;
}
struct TokTable {
    uint8_t *pool;   // token bytes; every token starts at a multiple of 4
    uint32_t *off;
    uint32_t *len;
    TokRec *rec;
    unsigned long long *vset; // open-addressing set keyed by the token bytes: id | (hash >> 32) << 32, ~0 = empty
    uint32_t vset_mask;
    uint32_t pool_cap;
};
constexpr unsigned long long VSET_EMPTY = ~0ull;

// Hash of a byte string, linear in the string so that the hash of a concatenation follows from the parts without their
// bytes: H(s) = sum (s[i] + 1) * B^(n-1-i) mod 2^64, H(xy) = H(x) * B^len(y) + H(y).  The selection creates the merged
// token from two records (rank, len, hash) it has already loaded; equality of byte strings is still decided by a full
// compare whenever hash and length agree.  Identical on host and device.
constexpr unsigned long long YB_HASH_B = 0x9E3779B97F4A7C15ull | 1ull;
YB_HD unsigned long long yb_hash_bytes(const uint8_t *p, uint32_t n) {
    unsigned long long h = 0;
    for (uint32_t i = 0; i < n; ++i) h = h * YB_HASH_B + (unsigned long long)(p[i] + 1u);
    return h;
}
YB_HD unsigned long long yb_hash_pow(uint32_t n) { // B^n
    unsigned long long r = 1, b = YB_HASH_B;
    while (n) {
        if (n & 1u) r *= b;
        b *= b;
        n >>= 1;
    }
    return r;
}
YB_HD unsigned long long yb_hash_concat(unsigned long long hx, unsigned long long hy, uint32_t ly) { return hx * yb_hash_pow(ly) + hy; }
YB_HD uint32_t yb_vset_home(unsigned long long h, uint32_t len) { // slot of a byte string (before masking)
    unsigned long long x = h ^ ((unsigned long long)len * 0xD1B54A32D192ED03ull);
    x ^= x >> 29;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 32;
    return (uint32_t)x;
}
YB_HD unsigned long long yb_vset_entry(uint32_t id, unsigned long long h) { return (unsigned long long)id | (h & 0xFFFFFFFF00000000ull); }

// byte i of the pool, past the caches (a token's bytes may have been written by another workgroup of this launch)
__device__ __forceinline__ uint32_t pool_byte_coherent(const uint8_t *pool, uint32_t i) {
    const uint32_t w = ld_coherent(reinterpret_cast<const uint32_t *>(pool) + (i >> 2));
    return (w >> ((i & 3u) * 8u)) & 0xffu;
}

// Python bytes order: unsigned bytewise, a proper prefix sorts lower.  Compares token t with the concatenation a + b
// (the token a merge has just created; its own bytes may not be in the pool yet).
__device__ __forceinline__ int tok_cmp_concat(const TokTable &tt, uint32_t t, uint32_t a, uint32_t b) {
    const uint8_t *pt = tt.pool + tt.off[t], *pa = tt.pool + tt.off[a], *pb = tt.pool + tt.off[b];
    const uint32_t lt = tt.len[t], la = tt.len[a], lc = la + tt.len[b];
    const uint32_t n = lt < lc ? lt : lc;
    for (uint32_t i = 0; i < n; ++i) {
        const int d = (int)pt[i] - (int)(i < la ? pa[i] : pb[i - la]);
        if (d) return d;
    }
    return (lt > lc) - (lt < lc);
}

// Python bytes order between two concatenations a1 + b1 and a2 + b2 (two tokens the same batch has just created)
__device__ __forceinline__ int tok_cmp_concat2(const TokTable &tt, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) {
    const uint8_t *pa1 = tt.pool + tt.off[a1], *pb1 = tt.pool + tt.off[b1], *pa2 = tt.pool + tt.off[a2], *pb2 = tt.pool + tt.off[b2];
    const uint32_t la1 = tt.len[a1], l1 = la1 + tt.len[b1], la2 = tt.len[a2], l2 = la2 + tt.len[b2];
    const uint32_t n = l1 < l2 ? l1 : l2;
    for (uint32_t i = 0; i < n; ++i) {
        const int d = (int)(i < la1 ? pa1[i] : pb1[i - la1]) - (int)(i < la2 ? pa2[i] : pb2[i - la2]);
        if (d) return d;
    }
    return (l1 > l2) - (l1 < l2);
}

// lexrank maintenance after the last selection created new tokens c_k = a_k + b_k (the batch in DevState; a_k, b_k are older
// tokens, their bytes are in the pool): a token moves up by one for every new token below it, and a new token's rank is the
// number of tokens below it.  The first block also writes the new tokens' bytes into the pool (the selection only reserved
// the place: it works from hashes, see select_body).
struct RankParams {
    TokTable tt;
    DevState *st;
};

__device__ __forceinline__ void rank_update_block(const RankParams &P, uint32_t block) {
    __shared__ uint32_t s_less[KMAX];
    __shared__ BatchMerge s_m[KMAX];
    DevState *st = P.st;
    if (st->done | st->halt) return;
    const uint32_t n = st->n_tokens, nb = min(st->n_batch, (uint32_t)KMAX);
    if (threadIdx.x < (uint32_t)KMAX) {
        s_less[threadIdx.x] = 0;
        s_m[threadIdx.x] = threadIdx.x < nb ? st->batch[threadIdx.x] : BatchMerge{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    uint32_t n_new = 0;
    for (uint32_t k = 0; k < nb; ++k) n_new += s_m[k].is_new;
    if (n_new == 0) return;
    const uint32_t n_old = n - n_new; // (new ids are the last ones: n_old .. n - 1 in batch order)
    if (block * BLOCK >= n_old) return;
    if (block == 0) {
        for (uint32_t k = 0; k < nb; ++k) { // the new tokens' bytes, four at a time (the pool is 4-byte granular), written through
            if (!s_m[k].is_new) continue;
            const uint32_t a = s_m[k].a, b = s_m[k].b, c = s_m[k].c;
            const uint8_t *pa = P.tt.pool + P.tt.off[a], *pb = P.tt.pool + P.tt.off[b];
            const uint32_t la = P.tt.len[a], lc = la + P.tt.len[b];
            uint32_t *dst = reinterpret_cast<uint32_t *>(P.tt.pool + P.tt.off[c]);
            for (uint32_t w = threadIdx.x; w * 4u < lc; w += BLOCK) {
                uint32_t v = 0;
                for (uint32_t q = 0; q < 4u; ++q) {
                    const uint32_t i = w * 4u + q;
                    if (i < lc) v |= (uint32_t)(i < la ? pa[i] : pb[i - la]) << (8u * q);
                }
                st_coherent(&dst[w], v);
            }
        }
        if (threadIdx.x < nb && s_m[threadIdx.x].is_new) { // new tokens among themselves
            const uint32_t k = threadIdx.x;
            uint32_t below = 0;
            for (uint32_t k2 = 0; k2 < nb; ++k2)
                if (k2 != k && s_m[k2].is_new && tok_cmp_concat2(P.tt, s_m[k2].a, s_m[k2].b, s_m[k].a, s_m[k].b) < 0) ++below;
            if (below) atomicAdd(&s_less[k], below);
        }
    }
    const uint32_t t = block * BLOCK + threadIdx.x;
    uint32_t up = 0, lessbits = 0;
    if (t < n_old) {
        for (uint32_t k = 0; k < nb; ++k) {
            if (!s_m[k].is_new) continue;
            if (tok_cmp_concat(P.tt, t, s_m[k].a, s_m[k].b) > 0) ++up; else lessbits |= 1u << k;
        }
        if (up) st_coherent(&P.tt.rec[t].rank, P.tt.rec[t].rank + up); // (read by the selection of the same launch in the fused form)
    }
    for (uint32_t k = 0; k < nb; ++k) {
        if (!s_m[k].is_new) continue; // uniform
        const unsigned long long m = __ballot((lessbits >> k) & 1u);
        if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_less[k], (uint32_t)__popcll(m));
    }
    __syncthreads();
    if (threadIdx.x < nb && s_m[threadIdx.x].is_new && s_less[threadIdx.x]) atomicAdd(&P.tt.rec[s_m[threadIdx.x].c].rank, s_less[threadIdx.x]);
}

__global__ __launch_bounds__(BLOCK) void k_rank_update(RankParams P) { rank_update_block(P, blockIdx.x); }

// ---------------------------------------------------------------- skip index: signatures and the scan that uses them
struct SigParams {
    const uint16_t *tiles;
    const uint32_t *tile_len;
    uint32_t n_tiles;
    unsigned long long *sig;
    uint32_t sig_stride;
};
__global__ __launch_bounds__(BLOCK) void k_build_sig(SigParams P) {
    constexpr int ROW = SIG_ROWS + 2; // (padding: the transposed read below walks the tiles)
    __shared__ unsigned long long s_sig[SIG_TILES][ROW];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_groups = (P.n_tiles + SIG_TILES - 1) / SIG_TILES;
    for (uint32_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint32_t base = grp * SIG_TILES;
        for (int w = threadIdx.x; w < SIG_TILES * ROW; w += BLOCK) (&s_sig[0][0])[w] = 0ull;
        __syncthreads();
        for (int k = wib; k < SIG_TILES; k += WPB) {
            const uint32_t tile = base + k;
            const uint32_t len = tile < P.n_tiles ? P.tile_len[tile] : 0u;
            if (!len) continue;
            unsigned long long *sg = s_sig[k];
            const TileRegs r = load_tile(P.tiles, tile, len, lane);
            const uint32_t b0 = __builtin_amdgcn_readfirstlane(r.vb.x);
            const uint32_t na = next_lane(r.va.x, b0);
            const uint32_t nb = next_lane(r.vb.x, PADPAD);
#pragma unroll
            for (int seg = 0; seg < 2; ++seg) {
                const uint4 &v = seg ? r.vb : r.va;
                const uint32_t nx = (seg ? nb : na) & 0xffffu;
                const uint32_t p0 = seg ? 512 + lane * 8 : lane * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t x = elem16(v, j);
                    const uint32_t y = j < 7 ? elem16(v, j + 1) : nx;
                    if (p0 + j < len && x < YB_PAD && y < YB_PAD) {
                        const SigHash H = sig_hash(yb_pairkey(x, y));
                        atomicOr(&sg[H.row], H.mask);
                    }
                }
            }
        }
        __syncthreads();
        // transposed store: 8 consecutive tiles of one row are 64 contiguous bytes
        for (int idx = threadIdx.x; idx < SIG_ROWS * SIG_TILES; idx += BLOCK) {
            const int row = idx / SIG_TILES, col = idx % SIG_TILES;
            if (base + col < P.n_tiles) P.sig[(size_t)row * P.sig_stride + base + col] = s_sig[col][row];
        }
        __syncthreads();
    }
}

// ================================================================ long words (> LMAX-1 tokens): one workgroup per word
struct LongParams {
    uint16_t *tok;
    const unsigned long long *off;
    uint32_t *len;
    const uint32_t *freq; // NULL: 1
    uint32_t n_long;
    PairTable out;
    DevState *st;
    unsigned long long *sig; // per-word signatures, same blocked Bloom filter and layout as the tiles' (sig[row * stride + word]); may be NULL
    uint32_t sig_stride;
    uint32_t group;          // words per workgroup of apply_long_block (0: BLOCK).  Few long words: small groups, so that the words a
                             // merge hits are rewritten by many workgroups side by side and not by a handful one after the other
};

__global__ __launch_bounds__(BLOCK) void k_count_long(LongParams P) {
    const uint32_t i = blockIdx.x;
    if (i >= P.n_long) return;
    const uint16_t *t = P.tok + P.off[i];
    const uint32_t len = P.len[i];
    const long long w = P.freq ? (long long)P.freq[i] : 1;
    for (uint32_t p = threadIdx.x; p + 1 < len; p += BLOCK) gt_add(P.out, P.st, yb_pairkey(t[p], t[p + 1]), w);
}

// signatures of the long words (at load): one workgroup per word
__global__ __launch_bounds__(BLOCK) void k_build_sig_long(LongParams P) {
    __shared__ unsigned long long s_sig[SIG_ROWS];
    const uint32_t i = blockIdx.x;
    if (i >= P.n_long) return;
    for (int w = threadIdx.x; w < SIG_ROWS; w += BLOCK) s_sig[w] = 0ull;
    __syncthreads();
    const uint16_t *t = P.tok + P.off[i];
    const uint32_t len = P.len[i];
    for (uint32_t p = threadIdx.x; p + 1 < len; p += BLOCK) {
        const SigHash H = sig_hash(yb_pairkey(t[p], t[p + 1]));
        atomicOr(&s_sig[H.row], H.mask);
    }
    __syncthreads();
    for (int w = threadIdx.x; w < SIG_ROWS; w += BLOCK) P.sig[(size_t)w * P.sig_stride + i] = s_sig[w];
}

// The merge applied to the long words.  A workgroup takes BLOCK consecutive words: one thread tests one word's signature
// (ONE 8-byte load per word -- a merge whose pair occurs in no long word costs a launch of n_long / 256 workgroups that
// read 8 bytes per word and leave), the words that pass are rewritten one after the other by the whole workgroup
// (sequential greedy rewrite in chunks, thread 0 carries the state across chunks; the pairs it creates set their bits).
__device__ __forceinline__ void apply_long_block(const LongParams &P, uint32_t block) { // BLOCK threads; block = which BLOCK words
    __shared__ uint16_t s_in[LONG_CH + 4];
    __shared__ uint16_t s_o[LONG_CH];
    __shared__ uint32_t s_j, s_o_pos, s_adv, s_nout, s_nhit, s_sites, s_prev[3], s_wsum[WPB];
    __shared__ uint32_t s_hit[BLOCK];
    DevState *st = P.st;
    if (st->done | st->halt) return;
    const uint32_t nbatch = min(st->n_batch, (uint32_t)KMAX);
    for (uint32_t bk = 0; bk < nbatch; ++bk) { // the merges of the batch, in order (a workgroup owns its words)
    const uint32_t a = st->batch[bk].a, b = st->batch[bk].b, c = st->batch[bk].c;
    const uint32_t self = yb_pairkey(a, b);
    __syncthreads();
    if (threadIdx.x == 0) s_nhit = 0;
    __syncthreads();
    {
        const uint32_t group = P.group ? min(P.group, (uint32_t)BLOCK) : (uint32_t)BLOCK;
        const uint32_t i = block * group + threadIdx.x;
        bool maybe = threadIdx.x < group && i < P.n_long;
        if (maybe && P.sig) {
            const SigHash H = sig_hash(self);
            maybe = (P.sig[(size_t)H.row * P.sig_stride + i] & H.mask) == H.mask;
        }
        if (maybe) s_hit[atomicAdd(&s_nhit, 1u)] = i;
    }
    __syncthreads();
    const uint32_t n_hit = s_nhit;
    for (uint32_t hidx = 0; hidx < n_hit; ++hidx) {
    const uint32_t i = s_hit[hidx];
    uint16_t *t = P.tok + P.off[i];
    const uint32_t len = P.len[i];
    const long long w = P.freq ? (long long)P.freq[i] : 1;
    int any = 0;
    for (uint32_t p = threadIdx.x; p + 1 < len; p += BLOCK) any |= (t[p] == a) & (t[p + 1] == b);
    if (!__syncthreads_or(any)) continue;

    if (threadIdx.x == 0) {
        s_j = 0;
        s_o_pos = 0;
        s_sites = 0;
        s_prev[0] = s_prev[1] = s_prev[2] = 0; // what stands in front of the next chunk: its old and new token, and whether there is one
    }
    __syncthreads();
    while (true) {
        const uint32_t j = s_j;
        if (j >= len) break;
        const uint32_t n_in = min((uint32_t)(LONG_CH + 3), len - j);
        for (uint32_t q = threadIdx.x; q < n_in; q += BLOCK) s_in[q] = t[j + q];
        __syncthreads();
        if (a != b) {
            // Sites of a != b never overlap: every position decides by itself.  A thread owns four consecutive positions of the
            // chunk; a site's deltas need the staged tokens around it (tile_logic's rules: where two sites touch, the pair between
            // them belongs to the right one) and, at position 0, what the chunk before left behind.  One block-wide prefix sum of the
            // site counts gives every kept token its place.  (The sequential form -- one thread walking the chunk -- cost ~12 us
            // per word and merge: 20,000 words of 64..300 random letters made a merge of the 256 MiB job 3.6 x as expensive.)
            static_assert(LONG_CH == 4 * BLOCK, "a thread owns four positions of a chunk");
            const uint32_t avail = len - j;                           // input tokens from j on
            const uint32_t n_body = min((uint32_t)LONG_CH, avail);    // positions this chunk decides
            auto site = [&](int q) -> bool { return q >= 0 && (uint32_t)q < n_body && (uint32_t)q + 1u < avail && s_in[q] == a && s_in[q + 1] == b; };
            const int q0 = (int)threadIdx.x * 4;
            uint32_t fl = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) fl |= (uint32_t)site(q0 + u) << u;
            const uint32_t mine = __popc(fl);
            const uint32_t inc = wave_inclusive_sum(mine);
            if ((threadIdx.x & 63) == 63) s_wsum[threadIdx.x >> 6] = inc;
            __syncthreads();
            uint32_t before = inc - mine, total = 0;
#pragma unroll
            for (int wv = 0; wv < WPB; ++wv) {
                if (wv < (int)(threadIdx.x >> 6)) before += s_wsum[wv];
                total += s_wsum[wv];
            }
            const bool last_is_site = site((int)n_body - 1);           // it consumes the first token of the next chunk too
            const uint32_t n_cons = n_body + (last_is_site ? 1u : 0u);
            uint32_t sb = before;                                       // sites in front of position q0 + u
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = q0 + u;
                if ((uint32_t)q >= n_body) break;
                const bool f = (fl >> u) & 1u;
                const bool second = site(q - 1);                        // the b of the site in front: leaves
                if (!second) s_o[(uint32_t)q - sb] = (uint16_t)(f ? c : s_in[q]);
                if (f) {
                    // left neighbour: the chunk before (position 0), the site that ends at q - 1, or the token at q - 1
                    bool have_prev = true;
                    uint32_t prev_old, prev_new;
                    if (q == 0) {
                        have_prev = s_prev[2] != 0u;
                        prev_old = s_prev[0];
                        prev_new = s_prev[1];
                    } else if (site(q - 2)) {
                        prev_old = b;
                        prev_new = c;
                    } else {
                        prev_old = prev_new = s_in[q - 1];
                    }
                    // (deltas to the merged pair's own key are skipped: the selection set its count to 0)
                    if (have_prev) {
                        if (yb_pairkey(prev_old, a) != self) gt_add(P.out, st, yb_pairkey(prev_old, a), -w);
                        gt_add(P.out, st, yb_pairkey(prev_new, c), +w);
                        if (P.sig) sig_set_pair(P.sig, P.sig_stride, i, yb_pairkey(prev_new, c));
                    }
                    if ((uint32_t)q + 2u < avail) {
                        const uint32_t y = s_in[q + 2];
                        const bool next_site = (uint32_t)q + 3u < avail && y == a && s_in[q + 3] == b;
                        if (!next_site) {
                            if (yb_pairkey(b, y) != self) gt_add(P.out, st, yb_pairkey(b, y), -w);
                            gt_add(P.out, st, yb_pairkey(c, y), +w);
                            if (P.sig) sig_set_pair(P.sig, P.sig_stride, i, yb_pairkey(c, y));
                        }
                    }
                    ++sb;
                }
            }
            __syncthreads(); // (s_prev has been read)
            if (threadIdx.x == 0) {
                s_adv = n_cons;
                s_nout = n_cons - total; // (every site removes one of the tokens it consumes)
                s_sites += total;
                // what the next chunk finds in front of it
                const int qe = (int)n_cons - 1;
                if (site(qe - 1)) {
                    s_prev[0] = b;
                    s_prev[1] = c;
                } else {
                    s_prev[0] = s_prev[1] = s_in[qe];
                }
                s_prev[2] = 1u;
            }
        } else if (threadIdx.x == 0) {
            // a == b: the greedy parity rule over runs of a (trainer.py:276-285) is sequential: one thread walks the chunk
            bool have_prev = s_prev[2] != 0u;
            uint32_t prev_old = s_prev[0], prev_new = s_prev[1];
            uint32_t q = 0, no = 0, ns = 0;
            while (q < (uint32_t)LONG_CH && j + q < len) {
                if (j + q + 1 < len && s_in[q] == a && s_in[q + 1] == b) {
                    // (deltas to the merged pair's own key are skipped: k_select set its count to 0)
                    if (have_prev) {
                        if (yb_pairkey(prev_old, a) != self) gt_add(P.out, st, yb_pairkey(prev_old, a), -w);
                        gt_add(P.out, st, yb_pairkey(prev_new, c), +w);
                        if (P.sig) sig_set_pair(P.sig, P.sig_stride, i, yb_pairkey(prev_new, c));
                    }
                    if (j + q + 2 < len) {
                        bool next_site = (j + q + 3 < len) && s_in[q + 2] == a && s_in[q + 3] == b;
                        if (!next_site) {
                            if (yb_pairkey(b, s_in[q + 2]) != self) gt_add(P.out, st, yb_pairkey(b, s_in[q + 2]), -w);
                            gt_add(P.out, st, yb_pairkey(c, s_in[q + 2]), +w);
                            if (P.sig) sig_set_pair(P.sig, P.sig_stride, i, yb_pairkey(c, s_in[q + 2]));
                        }
                    }
                    s_o[no++] = (uint16_t)c;
                    prev_old = b;
                    prev_new = c;
                    have_prev = true;
                    q += 2;
                    ns++;
                } else {
                    uint16_t x = s_in[q];
                    s_o[no++] = x;
                    prev_old = prev_new = x;
                    have_prev = true;
                    q += 1;
                }
            }
            s_adv = q;
            s_nout = no;
            s_sites += ns;
            s_prev[0] = prev_old;
            s_prev[1] = prev_new;
            s_prev[2] = have_prev ? 1u : 0u;
        }
        __syncthreads();
        const uint32_t o0 = s_o_pos, n_out = s_nout;
        for (uint32_t q = threadIdx.x; q < n_out; q += BLOCK) t[o0 + q] = s_o[q];
        __syncthreads();
        if (threadIdx.x == 0) {
            s_j = j + s_adv;
            s_o_pos = o0 + n_out;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        P.len[i] = s_o_pos;
        if (s_sites) atomicAdd(&st->sites, (unsigned long long)s_sites);
    }
    __syncthreads();
    }
    }
}
__global__ __launch_bounds__(BLOCK) void k_apply_long(LongParams P) { apply_long_block(P, blockIdx.x); }

// lengths of the long words (bytes = tokens at load), for the prefix sum that places them in their buffer
__global__ void k_long_lengths(const unsigned long long *off, const uint32_t *long_word, uint32_t n_long, uint32_t *out_len, uint32_t *too_long) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_long) return;
    const unsigned long long L = off[long_word[i] + 1] - off[long_word[i]];
    if (L > 0xFFFFFFFFull) atomicExch(too_long, 1u);
    out_len[i] = (uint32_t)L;
}

// ================================================================ argmax (trainer.py:246)
__device__ __forceinline__ bool best_gt(const Best &x, const Best &y) {
    return x.cnt > y.cnt || (x.cnt == y.cnt && x.rk > y.rk);
}

// Maximum of (cnt, rk) over the wave with DPP moves only (six steps of the row_shr / row_bcast ladder leave it in lane 63;
// a lane without a source keeps its own value).  rk is unique per pair, so the winner's lane is found with one ballot and
// its key / slot are read out of it.  Returns the same record in every lane.
__device__ __forceinline__ Best best_wave_reduce(Best v) {
    uint32_t hi = (uint32_t)(v.cnt >> 32), lo = (uint32_t)v.cnt, rk = v.rk;
#define YB_BEST_STEP(ctrl, rowmask)                                                                          \
    {                                                                                                        \
        const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, ctrl, rowmask, 0xf, false); \
        const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, ctrl, rowmask, 0xf, false); \
        const uint32_t r2 = (uint32_t)__builtin_amdgcn_update_dpp((int)rk, (int)rk, ctrl, rowmask, 0xf, false); \
        const bool g = h2 > hi || (h2 == hi && (l2 > lo || (l2 == lo && r2 > rk)));                          \
        hi = g ? h2 : hi;                                                                                    \
        lo = g ? l2 : lo;                                                                                    \
        rk = g ? r2 : rk;                                                                                    \
    }
    YB_BEST_STEP(0x111, 0xf) // row_shr:1
    YB_BEST_STEP(0x112, 0xf) // row_shr:2
    YB_BEST_STEP(0x114, 0xf) // row_shr:4
    YB_BEST_STEP(0x118, 0xf) // row_shr:8
    YB_BEST_STEP(0x142, 0xa) // row_bcast:15 -> rows 1, 3
    YB_BEST_STEP(0x143, 0xc) // row_bcast:31 -> rows 2, 3
#undef YB_BEST_STEP
    const uint32_t mh = __builtin_amdgcn_readlane(hi, 63), ml = __builtin_amdgcn_readlane(lo, 63);
    const uint32_t mr = __builtin_amdgcn_readlane(rk, 63);
    const unsigned long long mc = ((unsigned long long)mh << 32) | ml;
    const unsigned long long who = __ballot(v.cnt == mc && v.rk == mr);
    const int src = __ffsll((long long)who) - 1; // never empty: the maximum is some lane's value
    Best r;
    r.cnt = mc;
    r.rk = mr;
    r.key = __builtin_amdgcn_readlane(v.key, src);
    r.slot = __builtin_amdgcn_readlane(v.slot, src);
    r.pad = 0;
    return r;
}

// Sum of a 64-bit value over the wave, same ladder (a lane without a source adds 0); every lane gets the total.
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long x) {
    uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
#define YB_SUM_STEP(ctrl, rowmask, bc)                                                                     \
    {                                                                                                      \
        const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, ctrl, rowmask, 0xf, bc);      \
        const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, ctrl, rowmask, 0xf, bc);      \
        const unsigned long long t = (((unsigned long long)hi << 32) | lo) + (((unsigned long long)h2 << 32) | l2); \
        lo = (uint32_t)t;                                                                                  \
        hi = (uint32_t)(t >> 32);                                                                          \
    }
    YB_SUM_STEP(0x111, 0xf, true)
    YB_SUM_STEP(0x112, 0xf, true)
    YB_SUM_STEP(0x114, 0xf, true)
    YB_SUM_STEP(0x118, 0xf, true)
    YB_SUM_STEP(0x142, 0xa, false)
    YB_SUM_STEP(0x143, 0xc, false)
#undef YB_SUM_STEP
    return ((unsigned long long)__builtin_amdgcn_readlane(hi, 63) << 32) | __builtin_amdgcn_readlane(lo, 63);
}

struct ArgmaxParams {
    PairTable table;
    const TokRec *rec;
    Best *partials; // one per block
    DevState *st;
};

__global__ __launch_bounds__(BLOCK) void k_argmax_partial(ArgmaxParams P) {
    __shared__ Best s_b[WPB];
    if (P.st->done | P.st->halt) return;
    Best best{0ull, 0u, EMPTY, 0u, 0u};
    const uint32_t cap = P.table.cap;
    for (uint32_t s = blockIdx.x * BLOCK + threadIdx.x; s < cap; s += gridDim.x * BLOCK) {
        const long long cn = (long long)P.table.cnt[s]; // empty slots hold 0: the key is read for candidates only
        if (cn <= 0 || (unsigned long long)cn < best.cnt) continue;
        const uint32_t k = P.table.keys[s];
        if (k == EMPTY) continue;
        Best e{(unsigned long long)cn, (P.rec[k >> 16].rank << 16) | P.rec[k & 0xffffu].rank, k, s, 0u};
        if (best_gt(e, best)) best = e;
    }
    best = best_wave_reduce(best);
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    if (lane == 0) s_b[wib] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < WPB; ++i)
            if (best_gt(s_b[i], best)) best = s_b[i];
        P.partials[blockIdx.x] = best;
    }
}

struct SelectParams {
    const Best *partials;
    uint32_t n_partials;
    TokTable tt;
    DevState *st;
    // per-merge records
    uint32_t *rec_left, *rec_right, *rec_merged;
    unsigned long long *rec_count;
    unsigned long long *rec_sites;      // sites merged by iteration i (written when iteration i+1 is selected)
    unsigned long long *rec_live_slots; // live slots read by iteration i's apply pass
    uint32_t rec_base;                  // iter value at the start of this yabpe_train call
    PairTable table;                    // the selected pair's entry is zeroed here
    DeltaHdr *delta_hdr;                // multi-GPU: this rank's send header (count reset here), else NULL
    unsigned long long *blk_stats;      // per-workgroup counters of the last k_apply
    uint32_t n_blk;
    CandState *cs;                      // != NULL: the partials come from k_argmax_cand
    uint32_t flat_single;               // 1: flat layout on one GPU -- a merge's count is the number of its sites (a != b): per-merge site log of a batch
};

// Adds the per-workgroup counters of the last apply pass to DevState and clears them (one workgroup).
__device__ __forceinline__ void fold_block_stats(DevState *st, unsigned long long *blk, uint32_t n_blk) {
    __shared__ unsigned long long s_fold[2];
    if (threadIdx.x < 2) s_fold[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long a = 0, f = 0;
    for (uint32_t i = threadIdx.x; i < n_blk; i += BLOCK) {
        a += blk[2 * i];
        f += blk[2 * i + 1];
        blk[2 * i] = 0;
        blk[2 * i + 1] = 0;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        f += __shfl_xor(f, o);
    }
    if ((threadIdx.x & 63) == 0 && (a | f)) {
        atomicAdd(&s_fold[0], a);
        atomicAdd(&s_fold[1], f);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st->sites += s_fold[0];
        st->live_slots -= s_fold[1];
    }
    __syncthreads();
}

struct FoldParams {
    DevState *st;
    unsigned long long *blk_stats;
    uint32_t n_blk;
};
__global__ __launch_bounds__(BLOCK) void k_fold_stats(FoldParams P) { fold_block_stats(P.st, P.blk_stats, P.n_blk); }


#ifdef YB_PROFILE_SCAN
__device__ unsigned long long g_stop_hist[65536]; // per selection (index: DevState::iter when it ran): why the batch ended | batch size << 8 | window entries << 16
#define YB_STOP(it, why, n, nw) do { if (lane == 0) g_stop_hist[(it) & 0xFFFFu] = (unsigned long long)(why) | ((unsigned long long)(n) << 8) | ((unsigned long long)(nw) << 16); } while (0)
__device__ unsigned long long g_launch_prof[65536 * 4]; // per merge: min start, max end of the workgroups, -, selection end
__device__ unsigned long long g_sel_prof[16];
__device__ unsigned long long g_sel_acc[3 * 24]; // three ranges of merges x (count, sum of stamp i - stamp 8 ..., [12] merges selected, [13] window entries)
#define YB_SEL_STAMP(i) do { if (threadIdx.x == 0) g_sel_prof[i] = wall_clock64(); } while (0)
#else
#define YB_SEL_STAMP(i) do { } while (0)
#define YB_STOP(it, why, n, nw) do { } while (0)
#endif
// What a thread knows about its best candidate: the argmax record plus the two tokens' (len, hash), so that the winner's
// merged token needs no further trip to the token table.
struct BestEx {
    Best b;
    unsigned long long hx, hy;
    uint32_t lx, ly;
    uint32_t have; // hx .. ly are valid
};
__device__ __forceinline__ BestEx best_ex_none() { return BestEx{Best{0ull, 0u, EMPTY, 0u, 0u}, 0ull, 0ull, 0u, 0u, 0u}; }

// Best entry among list[first, first + step, ...), four entries per thread in flight: count, ranks, lengths and hashes of
// an entry are one round trip (the key rides in the list, a token's record is 16 B).  Every load goes past the caches: in
// the fused form the list, the counts and the ranks were written by other workgroups of the same launch.
__device__ __forceinline__ void ld_rec_coherent(const TokRec *r, uint32_t &rank, uint32_t &len, unsigned long long &hash) {
    const unsigned long long *q = reinterpret_cast<const unsigned long long *>(r);
    const unsigned long long a = ld_coherent(q);
    hash = ld_coherent(q + 1);
    rank = (uint32_t)a;
    len = (uint32_t)(a >> 32);
}
template <int NF = 4> // entries in flight per thread
__device__ __forceinline__ BestEx cand_list_best(const PairTable &t, const TokRec *rec, uint32_t n, uint32_t first, uint32_t step) {
    BestEx best = best_ex_none();
    for (uint32_t i0 = first; i0 < n; i0 += (uint32_t)NF * step) {
        unsigned long long e[NF], cn[NF], hx[NF], hy[NF];
        uint32_t rl[NF], rr[NF], lx[NF], ly[NF];
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            const uint32_t i = i0 + (uint32_t)k * step;
            e[k] = i < n ? ld_coherent(&t.cand_list[i]) : ~0ull;
        }
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            cn[k] = 0ull;
            if (i0 + (uint32_t)k * step < n) {
                const uint32_t key = (uint32_t)(e[k] >> 32);
                cn[k] = ld_coherent(pt_count_ptr(t, e[k]));
                ld_rec_coherent(&rec[key >> 16], rl[k], lx[k], hx[k]);
                ld_rec_coherent(&rec[key & 0xffffu], rr[k], ly[k], hy[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < NF; ++k) {
            if ((long long)cn[k] <= 0) continue;
            const Best b{cn[k], (rl[k] << 16) | rr[k], (uint32_t)(e[k] >> 32), (uint32_t)e[k], 0u};
            if (best_gt(b, best.b)) best = BestEx{b, hx[k], hy[k], lx[k], ly[k], 1u};
        }
    }
    return best;
}

// ---------------------------------------------------------------- batches of merges (sparse phase)
// Sequential BPE (trainer.py:241-300) selects merge t+1 from the counts merge t left behind.  One look at the pair table
// can nevertheless fix SEVERAL consecutive merges, whenever it proves that applying the first ones cannot change which pair
// comes next.  Walk the pairs in selection order -- (count, bytes(left), bytes(right)) descending -- and accept candidate
// j = (p, q) with count n_j after the accepted merges i = (a_i, b_i) -> c_i iff
//   (1) q != a_i and p != b_i for every i: merge i lowers only the counts of (x, a_i), (b_i, y) and (a_i, b_i) itself
//       (trainer.py:264-273), so j's count is still n_j when its turn comes;
//   (2) no pair that merge i CREATES can come before j.  Such a pair contains c_i -- (x, c_i), (c_i, y), (c_i, c_i') -- and
//       its count is at most what the pair it grew out of had: count(x, a_i), count(b_i, y), count(b_i, a_i').  A pair with a
//       larger count than n_j would have been walked before j and ended the batch by rule (1), so only pairs that TIE with j
//       matter: for every pair e = (l, r) with count == n_j that is not in the batch and has r == a_i or l == b_i, each pair it
//       can turn into must sort below j by bytes.  Old tokens compare by lexrank; a new token c_i = a_i + b_i by its first
//       eight bytes and its length (TokRec::pre8), and "cannot tell" counts as "comes before j";
//   (3) c_i is a NEW token (an existing token's pairs already have counts) and a_i != b_i (a run a a a leaves (c, a) pairs
//       whose count is bounded only by the merged pair's own); such a merge may close a batch but not sit inside one;
// and stop at the first candidate that fails.  Every pair with count >= T is on the candidate list, so all of this is
// decided from what the selection reads anyway.  tools/batch_sim.cpp replays the rule against sequential BPE on the CPU.
constexpr int WIN = 128; // window of the walk: the candidate-list entries with the highest counts (whole count levels)
struct WinEnt {
    unsigned long long cnt;
    uint32_t key, slot, rk, lx, ly, pad; // lx, ly: byte lengths of the two tokens
    unsigned long long hx, hy, px, py;   // their hashes and 8-byte prefixes
};
struct AccEnt { // an accepted merge: what the tie rule needs about it
    uint32_t a, b, lc, win; // lc: byte length of the merged token, win: its window entry
    unsigned long long pc;  // 8-byte prefix of the merged token
};
// first eight bytes of x + y (big-endian, zero padded), from the parts' prefixes
__device__ __forceinline__ unsigned long long pre8_concat(unsigned long long px, uint32_t lx, unsigned long long py) {
    return lx >= 8u ? px : (px | (py >> (8u * lx)));
}
// Python bytes order from prefixes and lengths: -1 / 0 / +1, or 2 when the first eight bytes do not decide
__device__ __forceinline__ int cmp_pre8(unsigned long long px, uint32_t lx, unsigned long long py, uint32_t ly) {
    if (px != py) return px < py ? -1 : 1; // (a zero pad byte sorts like a string that has ended: a proper prefix is smaller)
    if (lx <= 8u && ly <= 8u) return lx < ly ? -1 : lx > ly ? 1 : 0;
    return 2;
}

// The selection as a launch of its own (k_select, or the last workgroup of k_argmax_cand): one workgroup commits a pending
// halt, folds the apply pass's counters, reduces the argmax partials, applies the stop rules and creates the merged token.
// ONE merge (DevState::batch[0]).  Used when no merge is pending (start of a job, after a halt) and when no candidate list
// can prove the maximum; the per-batch launches end with select_eval below instead.
//   - thread 0 keeps the DevState fields it needs in registers and writes back what it changes;
//   - the merged token is created from the two tokens' records: its hash follows from theirs (yb_hash_concat), one probe of
//     the byte-string set says whether those bytes are already a token (trainer.py:298), and its bytes are written by the
//     NEXT launch (rank_update_block) -- the winner's bytes are never read here unless hash and length match an entry.
__device__ __forceinline__ void select_body(const SelectParams &P) {
    __shared__ Best s_b[WPB];
    __shared__ uint32_t s_flag, s_slot, s_eq, s_lx, s_ly;
    __shared__ unsigned long long s_fold[2], s_hx, s_hy, s_px, s_py, s_ent;
    DevState *st = P.st;
    const int tid = threadIdx.x;
    YB_SEL_STAMP(1);
    uint32_t d_iter = 0, d_done = 0, d_halt = 0, d_halt_req = 0, d_n_tokens = 0, d_pool_used = 0, d_num_merges = 0;
    unsigned long long d_min_freq = 0, d_table_entries = 0, d_live_slots = 0, d_sites = 0, d_tokens_now = 0, d_prev_others = 0;
    unsigned long long candT = 0;
    uint32_t cand_over = 0, cand_n = 0;
    if (tid == 0) {
        d_iter = st->iter;
        d_done = st->done;
        d_halt = st->halt;
        d_n_tokens = st->n_tokens;
        d_pool_used = st->pool_used;
        d_num_merges = st->num_merges;
        d_min_freq = st->min_freq;
        d_live_slots = st->live_slots;
        d_tokens_now = st->tokens_now;
        d_prev_others = st->batch_others;
        // fields other workgroups of this launch may have moved (atomics): read them past the caches
        d_halt_req = ld_coherent(&st->halt_req);
        d_table_entries = ld_coherent(&st->table_entries);
        d_sites = ld_coherent(&st->sites);
        if (P.cs) {
            candT = P.cs->T;
            cand_over = ld_coherent(&P.cs->overflow);
            cand_n = ld_coherent(&P.cs->n);
        }
        s_fold[0] = 0;
        s_fold[1] = 0;
        s_flag = 0;
    }
    // the counters of the last apply pass (one slot per workgroup): summed and cleared
    unsigned long long fa = 0, ff = 0;
    {
        unsigned long long *bs = P.blk_stats;
        for (uint32_t i = tid; i < P.n_blk; i += BLOCK) {
            const unsigned long long x = ld_coherent(&bs[2 * i]), y = ld_coherent(&bs[2 * i + 1]);
            fa += x;
            ff += y;
            if (x | y) {
                bs[2 * i] = 0ull;
                bs[2 * i + 1] = 0ull;
            }
        }
    }
    Best best{0ull, 0u, EMPTY, 0u, 0u};
    for (uint32_t i = tid; i < P.n_partials; i += BLOCK) {
        const Best pe = best_load_coherent(&P.partials[i]);
        if (best_gt(pe, best)) best = pe;
    }
    fa = wave_sum_u64(fa);
    ff = wave_sum_u64(ff);
    best = best_wave_reduce(best);
    __syncthreads(); // s_fold zeroed
    if ((tid & 63) == 0) {
        s_b[tid >> 6] = best;
        if (fa | ff) {
            atomicAdd(&s_fold[0], fa);
            atomicAdd(&s_fold[1], ff);
        }
    }
    __syncthreads();
    YB_SEL_STAMP(2);
    if (tid == 0) {
        if (d_halt == 0 && d_halt_req != 0) d_halt = d_halt_req;
        if (P.delta_hdr) { // this rank's send header for the next exchange
            P.delta_hdr->count = 0ull;
            P.delta_hdr->halt = d_halt;
        }
        // Deterministic across ranks (all replicas hold the same keys): stop before the table gets crowded, so that
        // no replica can run out of probes on its own.
        if (d_halt == 0 && d_table_entries * 5ull > (unsigned long long)P.table.cap * 4ull) d_halt = HALT_TABLE_FULL; // > 80 % full
        d_sites += s_fold[0];
        d_live_slots -= s_fold[1];
        if (d_done | d_halt) {
            s_flag = 1;
        } else {
            for (int w = 1; w < WPB; ++w)
                if (best_gt(s_b[w], best)) best = s_b[w];
            s_b[0] = best;
            // close the log entries of the batch that has just been applied: merges [it - pb, it).  Flat layout, a != b: a
            // merge's sites are exactly its count, and only the last merge of a batch can have a == b -- it gets what is
            // left.  (Pooled words / several ranks: counts are not resident sites; the whole batch is logged on its last merge.)
            const uint32_t it = d_iter;
            if (it > P.rec_base && d_sites) { // (0: already closed, this is a re-run)
                // (the batch's other merges were logged with their counts when they were selected: what is left is the last one's)
                const unsigned long long others = P.flat_single ? d_prev_others : 0ull;
                P.rec_sites[it - 1 - P.rec_base] = others <= d_sites ? d_sites - others : d_sites;
            }
            d_tokens_now -= d_sites;
            d_sites = 0;
            if (P.cs && it < d_num_merges && (best.cnt < candT || cand_over)) {
                // the candidate set no longer proves that this is the maximum: the host redoes this merge with a full scan
                d_halt = HALT_RESCAN;
                s_flag = 1;
            } else if (it >= d_num_merges || best.cnt == 0 || best.cnt < d_min_freq) {
                // stop rules: iteration limit (trainer.py:241), no pairs (:242-243), min_frequency (:247-248)
                d_done = 1;
                s_flag = 1;
            }
        }
        // what every path writes back (the rest follows when a merge has been selected)
        st->halt = d_halt;
        st->done = d_done;
        st->sites = d_sites;
        st->live_slots = d_live_slots;
        st->tokens_now = d_tokens_now;
        st->cand_n = cand_n;
        if (s_flag) st->n_batch = 0u;
    }
    __syncthreads();
    if (s_flag) return;
    YB_SEL_STAMP(3);
    const Best win = s_b[0];
    const uint32_t x = win.key >> 16, y = win.key & 0xffffu;
    if (tid < 2) { // the two tokens' records
        const TokRec r = P.tt.rec[tid ? y : x];
        if (tid) { s_hy = r.hash; s_ly = r.len; s_py = r.pre8; } else { s_hx = r.hash; s_lx = r.len; s_px = r.pre8; }
    }
    __syncthreads();
    YB_SEL_STAMP(4);
    const uint32_t lx = s_lx, L = s_lx + s_ly, pu = d_pool_used; // (thread 0 only)
    const unsigned long long H = yb_hash_concat(s_hx, s_hy, s_ly);
    // "merged not in vocab" (trainer.py:298): probe the byte-string set; entries carry 32 bits of the hash, so a slot that
    // holds another string is passed over without looking at that string
    if (tid == 0) {
        uint32_t slot = yb_vset_home(H, L) & P.tt.vset_mask;
        unsigned long long ve = P.tt.vset[slot];
        while (ve != VSET_EMPTY && (ve >> 32) != (H >> 32)) {
            slot = (slot + 1) & P.tt.vset_mask;
            ve = P.tt.vset[slot];
        }
        s_slot = slot;
        s_ent = ve;
    }
    __syncthreads();
    YB_SEL_STAMP(5);
    uint32_t found = EMPTY;
    while (s_ent != VSET_EMPTY) { // same 32 hash bits: compare for real (rare: the token exists already, or a 2^-32 coincidence)
        const uint32_t cand = (uint32_t)s_ent;
        if (tid == 0) s_eq = 1;
        __syncthreads();
        const TokRec rc = P.tt.rec[cand];
        if (rc.len == L && rc.hash == H) {
            const uint32_t oc = P.tt.off[cand], ox = P.tt.off[x], oy = P.tt.off[y];
            int ne = 0;
            for (uint32_t i = tid; i < L; i += BLOCK)
                ne |= pool_byte_coherent(P.tt.pool, oc + i) != pool_byte_coherent(P.tt.pool, i < lx ? ox + i : oy + (i - lx));
            if (ne) s_eq = 0; // benign race: every writer stores 0
        } else if (tid == 0) {
            s_eq = 0;
        }
        __syncthreads();
        if (s_eq) {
            found = cand;
            break;
        }
        if (tid == 0) { // next slot with the same 32 hash bits, or the empty slot that ends the run
            uint32_t slot = (s_slot + 1) & P.tt.vset_mask;
            unsigned long long ve = P.tt.vset[slot];
            while (ve != VSET_EMPTY && (ve >> 32) != (H >> 32)) {
                slot = (slot + 1) & P.tt.vset_mask;
                ve = P.tt.vset[slot];
            }
            s_slot = slot;
            s_ent = ve;
        }
        __syncthreads();
    }
    YB_SEL_STAMP(6);
    if (tid == 0) {
        uint32_t cid = 0;
        uint32_t is_new = 0;
        bool ok = true;
        if (found != EMPTY) {
            cid = found; // bytes already a token: no id is consumed (trainer.py:298-300)
        } else if (d_n_tokens >= YB_MAX_TOKENS) {
            st->halt = HALT_VOCAB_FULL;
            st->n_batch = 0u;
            ok = false;
        } else if ((unsigned long long)pu + L + 4ull > P.tt.pool_cap) {
            st->halt = HALT_POOL_FULL;
            st->n_batch = 0u;
            ok = false;
        } else {
            cid = d_n_tokens; // merged = p0 + p1 (trainer.py:251): place reserved, bytes written by the next launch
            P.tt.off[cid] = pu;
            P.tt.len[cid] = L;
            P.tt.rec[cid] = TokRec{0u, L, H, pre8_concat(s_px, s_lx, s_py), 0ull};
            P.tt.vset[s_slot] = yb_vset_entry(cid, H);
            is_new = 1;
        }
        if (ok) {
            const uint32_t ri = d_iter - P.rec_base;
            P.rec_left[ri] = x; // merges.append(best_pair) (trainer.py:296)
            P.rec_right[ri] = y;
            P.rec_merged[ri] = cid;
            P.rec_count[ri] = win.cnt;
            P.rec_live_slots[ri] = d_live_slots;
            // after this merge no (x,y) adjacency is left anywhere (trainer.py:276-285), so its count is exactly 0:
            // set it here once instead of letting every workgroup subtract its share from one hot address
            P.table.cnt[win.slot] = 0ull;
            st->a = x;
            st->b = y;
            st->c = cid;
            st->best_count = win.cnt;
            st->c_is_new = is_new;
            st->batch[0] = BatchMerge{x, y, cid, is_new};
            st->n_batch = 1u;
            st->n_select = st->n_select + 1u;
            st->batch_others = 0ull;
            st->iter = d_iter + 1;
            st->pool_used = is_new ? ((pu + L + 3u) & ~3u) : pu;
            st->n_tokens = d_n_tokens + is_new;
#ifdef YB_PROFILE_LAUNCH
            g_launch_prof[(d_iter & 0xFFFFu) * 4 + 3] = wall_clock64(); // (same index as the launch that ran this selection: its st->iter at start)
#endif
        }
    }
    YB_SEL_STAMP(7);
}

// ---------------------------------------------------------------- the fused selection (tail of every per-batch launch)
// Same contract as select_body, for a BATCH of merges and built for the critical path: four round trips (list + state, counts, token records
// of the window, byte-string set probes) and, between them, one workgroup-wide maximum, the window (distinct pairs with
// count >= max - delta, LDS hash set) and then ONE WAVE that sorts the window, walks it under the batch rule with
// wave-level votes (no workgroup barriers), probes the byte-string set for every accepted merge side by side and commits.
// More than WIN pairs tying at the top, or a first merge whose bytes may exist already (a full byte compare decides), end in
// a batch of one.
// s_win: LDS for WIN window entries (8 KiB), the caller's -- a kernel whose own tables are dead by now lends them.
__device__ __forceinline__ void select_eval(const SelectParams &P, WinEnt *s_win) {
    __shared__ AccEnt s_acc[KMAX];
    __shared__ __attribute__((aligned(16))) uint32_t s_hset[2 * WIN]; // (the window's hash set; afterwards the packed sort keys live here)
    __shared__ unsigned long long s_red[WPB], s_fold[2];
    __shared__ uint32_t s_nwin;
    struct TopEnt { unsigned long long cnt; uint32_t key, win; }; // the entry at a position of the selection order
    __shared__ unsigned long long s_lim[2];
    __shared__ uint32_t s_flags[2];
    unsigned long long *s_pk = reinterpret_cast<unsigned long long *>(s_hset); // [WIN]
    __shared__ TopEnt s_top[KMAX];
    static_assert(BLOCK == 2 * WIN, "two threads rank one window entry");
    constexpr uint32_t HSET = 2 * WIN;
    DevState *st = P.st;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    YB_SEL_STAMP(1);
    // ---- round trip 1: the list's first 4 x BLOCK entries and its length, the DevState fields, the per-workgroup counters
    unsigned long long e[4], cn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = ld_coherent(&P.table.cand_list[tid + k * BLOCK]);
    const uint32_t n_list = min(ld_coherent(&P.cs->n), CAND_CAP);
    const uint32_t kmax = min(st->kmax, (uint32_t)KMAX);
    uint32_t d_iter = 0, d_done = 0, d_halt = 0, d_halt_req = 0, d_n_tokens = 0, d_pool_used = 0, d_num_merges = 0;
    unsigned long long d_min_freq = 0, d_table_entries = 0, d_live_slots = 0, d_sites = 0, d_tokens_now = 0, candT = 0, d_prev_others = 0;
    uint32_t cand_over = 0, cand_n = 0;
    if (tid == 0) {
        d_iter = st->iter;
        d_done = st->done;
        d_halt = st->halt;
        d_n_tokens = st->n_tokens;
        d_pool_used = st->pool_used;
        d_num_merges = st->num_merges;
        d_min_freq = st->min_freq;
        d_live_slots = st->live_slots;
        d_tokens_now = st->tokens_now;
        d_prev_others = st->batch_others;
        d_halt_req = ld_coherent(&st->halt_req);
        d_table_entries = ld_coherent(&st->table_entries);
        d_sites = ld_coherent(&st->sites);
        candT = P.cs->T;
        cand_over = ld_coherent(&P.cs->overflow);
        cand_n = ld_coherent(&P.cs->n);
        s_fold[0] = 0;
        s_fold[1] = 0;
        s_nwin = 0;
        s_flags[0] = 0u; // positions of the selection order that fail the batch rule
        s_flags[1] = 0u; // positions that hold a run merge (p == q)
        s_lim[0] = candT;
        s_lim[1] = d_min_freq;
    }
    unsigned long long vx[4], vy[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = k * BLOCK + tid;
        vx[k] = i < P.n_blk ? ld_coherent(&P.blk_stats[2 * i]) : 0ull;
        vy[k] = i < P.n_blk ? ld_coherent(&P.blk_stats[2 * i + 1]) : 0ull;
    }
    for (uint32_t hidx = tid; hidx < HSET; hidx += BLOCK) s_hset[hidx] = EMPTY;
    // ---- round trip 2: the counts (the slot rides in the list entry)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        cn[k] = 0ull;
        if ((uint32_t)(tid + k * BLOCK) < n_list) cn[k] = ld_coherent(pt_count_ptr(P.table, e[k]));
    }
    // (the counters are summed while the counts are on their way)
    unsigned long long fa = 0, ff = 0;
    {
        unsigned long long *bs = P.blk_stats;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t i = k * BLOCK + tid;
            fa += vx[k];
            ff += vy[k];
            if (i < P.n_blk && (vx[k] | vy[k])) {
                bs[2 * i] = 0ull;
                bs[2 * i + 1] = 0ull;
            }
        }
        for (uint32_t i = 4 * BLOCK + tid; i < P.n_blk; i += BLOCK) { // (grids beyond 1,024 workgroups: the streaming phase)
            const unsigned long long x = ld_coherent(&bs[2 * i]), y = ld_coherent(&bs[2 * i + 1]);
            fa += x;
            ff += y;
            if (x | y) {
                bs[2 * i] = 0ull;
                bs[2 * i + 1] = 0ull;
            }
        }
    }
    fa = wave_sum_u64(fa);
    ff = wave_sum_u64(ff);
    YB_SEL_STAMP(9);
    unsigned long long cmax = 0ull;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if ((long long)cn[k] <= 0) cn[k] = 0ull;
        cmax = cn[k] > cmax ? cn[k] : cmax;
    }
    for (uint32_t ti = 4u * BLOCK + tid; ti < n_list; ti += BLOCK) { // (a long list: rare, the host rebuilds it before)
        const long long c = (long long)ld_coherent(pt_count_ptr(P.table, ld_coherent(&P.table.cand_list[ti])));
        if (c > 0 && (unsigned long long)c > cmax) cmax = (unsigned long long)c;
    }
    YB_SEL_STAMP(10);
    cmax = best_wave_reduce(Best{cmax, 0u, 0u, 0u, 0u}).cnt;
    __syncthreads(); // (s_fold / s_nwin / s_hset initialised)
    if (lane == 0) {
        s_red[tid >> 6] = cmax;
        if (fa | ff) {
            atomicAdd(&s_fold[0], fa);
            atomicAdd(&s_fold[1], ff);
        }
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WPB; ++w) cmax = s_red[w] > cmax ? s_red[w] : cmax;
    YB_SEL_STAMP(12);
    // ---- the window: the DISTINCT pairs with count >= L = cmax - delta (whole count levels by construction; a pair may be
    // listed more than once, and how often differs between the replicas of a multi-GPU job: the set keeps repeats out, so every
    // rank walks the same window).  More than WIN of them: a quarter of the distance, again.
    unsigned long long L = 0ull;
    uint32_t n_win = 0;
    uint32_t wshift = min(max(st->win_shift, 1u), 20u), narrowed = 0;
    if (cmax) {
        unsigned long long delta = kmax > 1u ? min(max((unsigned long long)kmax, cmax >> wshift), 1ull << 31) : 0ull;
        for (int attempt = 0; attempt < 8; ++attempt) {
            L = cmax > delta ? cmax - delta : 1ull;
            // (never below the list's threshold: only pairs with count >= T are on EVERY replica's list -- a pair below it may be
            // listed on one rank and not on another, and the window's size decides how far a batch can go)
            if (kmax > 1u && L < s_lim[0] && cmax >= s_lim[0]) L = s_lim[0];
            auto put = [&](unsigned long long ent, unsigned long long c) {
                const uint32_t key = (uint32_t)(ent >> 32);
                uint32_t h = hash32(key) & (HSET - 1u);
                for (uint32_t probe = 0; probe < HSET; ++probe) {
                    const uint32_t was = atomicCAS(&s_hset[h], EMPTY, key);
                    if (was == key) return; // listed twice
                    if (was == EMPTY) {
                        const uint32_t idx = atomicAdd(&s_nwin, 1u);
                        if (idx < (uint32_t)WIN) {
                            s_win[idx].cnt = c;
                            s_win[idx].key = key;
                            s_win[idx].slot = (uint32_t)ent;
                        }
                        return;
                    }
                    h = (h + 1u) & (HSET - 1u);
                }
                s_nwin = WIN + 1u; // (the set is full: more distinct pairs than a window holds)
            };
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cn[k] >= L) put(e[k], cn[k]);
            for (uint32_t ti = 4u * BLOCK + tid; ti < n_list; ti += BLOCK) {
                const unsigned long long ent = ld_coherent(&P.table.cand_list[ti]);
                const long long c = (long long)ld_coherent(pt_count_ptr(P.table, ent));
                if (c > 0 && (unsigned long long)c >= L) put(ent, (unsigned long long)c);
            }
            __syncthreads();
            n_win = s_nwin;
            if (n_win <= (uint32_t)WIN) break;
            n_win = 0;
            if (delta == 0ull) break; // more than WIN pairs tie at the top: no window
            delta >>= 2;
            narrowed += 2;
            __syncthreads(); // (everybody has read s_nwin)
            for (uint32_t hidx = tid; hidx < HSET; hidx += BLOCK) s_hset[hidx] = EMPTY;
            if (tid == 0) s_nwin = 0;
            __syncthreads();
        }
    }
    if (cmax && n_win == 0) {
        // More than WIN distinct pairs tie at the top (small counts late in a small job): no walk -- the greatest of them by
        // (lexrank, lexrank) is the merge (trainer.py:246), found the plain way; it becomes a window of one.
        __shared__ Best s_tb[WPB];
        Best tb{0ull, 0u, EMPTY, 0u, 0u};
        auto consider = [&](unsigned long long ent) {
            const uint32_t key = (uint32_t)(ent >> 32);
            const uint32_t rl = (uint32_t)ld_coherent(reinterpret_cast<const unsigned long long *>(&P.tt.rec[key >> 16]));
            const uint32_t rr = (uint32_t)ld_coherent(reinterpret_cast<const unsigned long long *>(&P.tt.rec[key & 0xffffu]));
            const Best b{1ull, (rl << 16) | rr, key, (uint32_t)ent, 0u};
            if (best_gt(b, tb)) tb = b;
        };
#pragma unroll 1
        for (int k = 0; k < 4; ++k)
            if (cn[k] == cmax) consider(e[k]);
        for (uint32_t ti = 4u * BLOCK + tid; ti < n_list; ti += BLOCK) {
            const unsigned long long ent = ld_coherent(&P.table.cand_list[ti]);
            if (ld_coherent(pt_count_ptr(P.table, ent)) == cmax) consider(ent);
        }
        tb = best_wave_reduce(tb);
        __syncthreads(); // (the last read of s_nwin is behind everybody)
        if (lane == 0) s_tb[tid >> 6] = tb;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < WPB; ++w)
                if (best_gt(s_tb[w], tb)) tb = s_tb[w];
            s_win[0].cnt = cmax;
            s_win[0].key = tb.key;
            s_win[0].slot = tb.slot;
        }
        __syncthreads();
        n_win = 1;
    }
    YB_SEL_STAMP(13);
    // ---- round trip 3: the two token records of every window entry (one entry per thread)
    if ((uint32_t)tid < n_win) {
        const uint32_t key = s_win[tid].key;
        const unsigned long long *qx = reinterpret_cast<const unsigned long long *>(&P.tt.rec[key >> 16]);
        const unsigned long long *qy = reinterpret_cast<const unsigned long long *>(&P.tt.rec[key & 0xffffu]);
        const unsigned long long ax = ld_coherent(qx), hx = ld_coherent(qx + 1), px = ld_coherent(qx + 2);
        const unsigned long long ay = ld_coherent(qy), hy = ld_coherent(qy + 1), py = ld_coherent(qy + 2);
        s_win[tid].rk = ((uint32_t)ax << 16) | ((uint32_t)ay & 0xffffu);
        s_win[tid].lx = (uint32_t)(ax >> 32);
        s_win[tid].ly = (uint32_t)(ay >> 32);
        s_win[tid].hx = hx;
        s_win[tid].hy = hy;
        s_win[tid].px = px;
        s_win[tid].py = py;
        // (the sort key: count above the window's floor, then the lexranks -- count - L < 2^32: delta is capped)
        s_pk[tid] = ((s_win[tid].cnt - L) << 32) | (unsigned long long)(((uint32_t)ax << 16) | ((uint32_t)ay & 0xffffu));
    } else if (tid < WIN) {
        s_pk[tid] = 0ull; // (never above anything)
    }
    __syncthreads();
    YB_SEL_STAMP(11);
    // ---- the batch rule (above select_body), for every window entry side by side instead of a walk.  The entries in
    // selection order are the window sorted by (count, lexranks) descending -- distinct pairs: no ties -- and the merges
    // accepted before position j are exactly positions 0 .. j-1, so everything rule (1) and rule (2) ask about position j is
    // known without walking: an entry's two bit sets over the first KMAX positions (ra: bit i = "my right token is a_i",
    // lb: bit i = "my left token is b_i"), cut to the bits below j.  The batch is the prefix in front of the first position
    // that fails.  Position by counting: every entry against every other, packed keys in LDS (written with the records
    // above; two lanes of one wave per entry, 32 keys each, two keys per LDS read).
    const uint32_t wt = (uint32_t)(tid >> 6) * 32u + ((uint32_t)lane & 31u), whalf = (uint32_t)lane >> 5; // (BLOCK == 2 * WIN)
    uint32_t my_pos = 0xffffu, my_key = 0u;
    unsigned long long my_cnt = 0ull;
    {
        const unsigned long long mine = s_pk[wt];
        const ulonglong2 *pk2 = reinterpret_cast<const ulonglong2 *>(s_pk) + whalf * (uint32_t)(WIN / 4);
        uint32_t above = 0;
#pragma unroll 8
        for (uint32_t m = 0; m < (uint32_t)(WIN / 4); ++m) {
            const ulonglong2 kk = pk2[m];
            above += (kk.x > mine ? 1u : 0u) + (kk.y > mine ? 1u : 0u);
        }
        above += __shfl_xor(above, 32);
        if (whalf == 0u && wt < n_win) {
            my_pos = above;
            my_key = s_win[wt].key;
            my_cnt = s_win[wt].cnt;
            if (my_pos < (uint32_t)KMAX) {
                s_top[my_pos] = TopEnt{my_cnt, my_key, wt};
                s_acc[my_pos] = AccEnt{my_key >> 16, my_key & 0xffffu, s_win[wt].lx + s_win[wt].ly, wt, pre8_concat(s_win[wt].px, s_win[wt].lx, s_win[wt].py)};
            }
        }
    }
    __syncthreads();
    const uint32_t n_top = min(n_win, (uint32_t)KMAX);
    if (my_pos != 0xffffu) {
        const uint32_t topmask = (1u << n_top) - 1u;
        uint32_t ra = 0, lb = 0, tie = 0;
#pragma unroll
        for (uint32_t i2 = 0; i2 < (uint32_t)KMAX; ++i2) { // (positions >= n_top hold stale entries: masked below)
            const TopEnt te = s_top[i2];
            ra |= (uint32_t)((my_key & 0xffffu) == (te.key >> 16)) << i2;
            lb |= (uint32_t)((my_key >> 16) == (te.key & 0xffffu)) << i2;
            tie |= (uint32_t)(te.cnt == my_cnt) << i2;
        }
        ra &= topmask;
        lb &= topmask;
        if (my_pos < (uint32_t)KMAX) {
            const uint32_t below = (1u << my_pos) - 1u;
            // rule (1): q is some a_i or p some b_i of an earlier position; and the stop rules for a position behind the first
            if (my_pos != 0u && (((ra | lb) & below) != 0u || my_cnt < s_lim[0] || my_cnt < s_lim[1])) atomicOr(&s_flags[0], 1u << my_pos);
            if ((my_key >> 16) == (my_key & 0xffffu)) atomicOr(&s_flags[1], 1u << my_pos); // rule (3): a run merge closes the batch
        }
        // rule (2): I tie with position j, come after it, and hold a token a merge in front of j consumes: may I turn into a
        // pair that sorts in front of j?  Positions that qualify: tying, behind the first merge that touches one of my tokens,
        // in front of me.  (rare: only then are the records read)
        const uint32_t touch = ra | lb;
        uint32_t cand = touch ? tie & topmask & ~((2u << (__ffs((int)touch) - 1)) - 1u) & (my_pos < (uint32_t)KMAX ? (1u << my_pos) - 1u : ~0u) : 0u;
        for (; cand; cand &= cand - 1u) {
            const uint32_t j2 = (uint32_t)__ffs((int)cand) - 1u, below = (1u << j2) - 1u;
            const uint32_t cwin = s_top[j2].win;
            const unsigned long long c_px = s_win[cwin].px, c_py = s_win[cwin].py;
            const uint32_t c_lx = s_win[cwin].lx, c_ly = s_win[cwin].ly;
            bool left_new_blocks = false, right_new_ge = false;
            for (uint32_t m = lb & below; m; m &= m - 1u) { // (c_i, .) against (p, .)
                const AccEnt A = s_acc[__ffs((int)m) - 1];
                left_new_blocks |= cmp_pre8(A.pc, A.lc, c_px, c_lx) != -1;
            }
            for (uint32_t m = ra & below; m; m &= m - 1u) { // (., c_i) against (., q)
                const AccEnt A = s_acc[__ffs((int)m) - 1];
                right_new_ge |= cmp_pre8(A.pc, A.lc, c_py, c_ly) != -1;
            }
            const uint32_t rl = s_win[wt].rk >> 16, rp = s_win[cwin].rk >> 16; // lexranks of my left token and of p
            if (left_new_blocks || ((ra & below) != 0u && (rl > rp || (rl == rp && right_new_ge)))) atomicOr(&s_flags[0], 1u << j2);
        }
    }
    __syncthreads();
    YB_SEL_STAMP(14);
    if (tid >= 64) return; // ---- from here on: wave 0 alone, wave-level synchronisation only
    // ---- stop rules and the first merge (lane 0 holds the state)
    uint32_t flag = 0; // 1: nothing selected (done / halt)
    const unsigned long long best_cnt = cmax;
    if (lane == 0) {
        if (d_halt == 0 && d_halt_req != 0) d_halt = d_halt_req;
        if (P.delta_hdr) { // this rank's send header for the next exchange
            P.delta_hdr->count = 0ull;
            P.delta_hdr->halt = d_halt;
        }
        if (d_halt == 0 && d_table_entries * 5ull > (unsigned long long)P.table.cap * 4ull) d_halt = HALT_TABLE_FULL; // > 80 % full (the same on every rank)
        d_sites += s_fold[0];
        d_live_slots -= s_fold[1];
        if (d_done | d_halt) {
            flag = 1;
        } else {
            // close the log entries of the batch that has just been applied (see select_body)
            const uint32_t it = d_iter;
            if (it > P.rec_base && d_sites) {
                // (the batch's other merges were logged with their counts when they were selected: what is left is the last one's)
                const unsigned long long others = P.flat_single ? d_prev_others : 0ull;
                P.rec_sites[it - 1 - P.rec_base] = others <= d_sites ? d_sites - others : d_sites;
            }
            d_tokens_now -= d_sites;
            d_sites = 0;
            if (it < d_num_merges && (best_cnt < candT || cand_over)) {
                d_halt = HALT_RESCAN; // the candidate set no longer proves the maximum: the host redoes this merge with a full scan
                flag = 1;
            } else if (it >= d_num_merges || best_cnt == 0 || best_cnt < d_min_freq) {
                d_done = 1; // stop rules: iteration limit (trainer.py:241), no pairs (:242-243), min_frequency (:247-248)
                flag = 1;
            }
        }
        st->halt = d_halt;
        st->done = d_done;
        st->sites = d_sites;
        st->live_slots = d_live_slots;
        st->tokens_now = d_tokens_now;
        st->cand_n = cand_n;
        if (flag) st->n_batch = 0u;
        // next time: a narrower window after an overflow, a wider one when this one held few pairs
        if (kmax > 1u) st->win_shift = narrowed ? min(wshift + narrowed, 20u) : (n_win < 4u * kmax && wshift > 1u) ? wshift - 1u : wshift;
    }
    flag = __builtin_amdgcn_readfirstlane(flag);
    if (flag) return;
    YB_SEL_STAMP(3);
    // ---- the batch: positions [0, nacc) -- up to the first position that fails, the first run merge included, the limits
    const uint32_t it0 = __builtin_amdgcn_readfirstlane(d_iter), num_merges = __builtin_amdgcn_readfirstlane(d_num_merges);
    uint32_t nacc = min(min(kmax, num_merges - it0), n_top), why = 4u;
    (void)why;
    {
        const uint32_t fails = s_flags[0] & ~1u, runs = s_flags[1];
        if (nacc == n_top && n_top < kmax && n_top < num_merges - it0) why = 5u; // the window is walked through
        if (fails && (uint32_t)(__ffs((int)fails) - 1) < nacc) {
            nacc = (uint32_t)(__ffs((int)fails) - 1);
            why = 1u;
        }
        if (runs && (uint32_t)__ffs((int)runs) < nacc) {
            nacc = (uint32_t)__ffs((int)runs);
            why = 6u;
        }
    }
    YB_STOP(it0, why, nacc, n_win);
    // ---- round trip 4: the merged tokens: one probe of the byte-string set each, side by side (lane k: merge k)
    unsigned long long Hk = 0ull, ve = VSET_EMPTY;
    uint32_t Lk = 0, vslot = 0, wk = 0;
    if ((uint32_t)lane < nacc) {
        wk = s_acc[lane].win;
        const uint32_t lxk = s_win[wk].lx, lyk = s_win[wk].ly;
        Lk = lxk + lyk;
        Hk = yb_hash_concat(s_win[wk].hx, s_win[wk].hy, lyk);
        vslot = yb_vset_home(Hk, Lk) & P.tt.vset_mask;
        ve = ld_coherent(&P.tt.vset[vslot]);
        while (ve != VSET_EMPTY && (ve >> 32) != (Hk >> 32)) {
            vslot = (vslot + 1) & P.tt.vset_mask;
            ve = ld_coherent(&P.tt.vset[vslot]);
        }
    }
    YB_SEL_STAMP(5);
    const bool hit = (uint32_t)lane < nacc && ve != VSET_EMPTY; // same 32 hash bits as an existing token: a full compare decides
    // the batch ends in front of the first later merge whose token cannot be created blindly: possible hit, the same slot or the
    // same (hash, length) as an earlier merge of the batch, no id or no pool space left
    uint32_t bad = hit ? 1u : 0u;
    for (uint32_t k2 = 0; k2 + 1 < nacc; ++k2) {
        const uint32_t s2 = __builtin_amdgcn_readlane(vslot, k2), l2 = __builtin_amdgcn_readlane(Lk, k2);
        const unsigned long long h2 = ((unsigned long long)__builtin_amdgcn_readlane((uint32_t)(Hk >> 32), k2) << 32) | __builtin_amdgcn_readlane((uint32_t)Hk, k2);
        if ((uint32_t)lane < nacc && (uint32_t)lane > k2 && (s2 == vslot || (h2 == Hk && l2 == Lk))) bad = 1u;
    }
    // pool offsets: exclusive prefix of the 4-byte-rounded lengths (a handful of lanes)
    const uint32_t ntok0 = __builtin_amdgcn_readfirstlane(d_n_tokens), pool0 = __builtin_amdgcn_readfirstlane(d_pool_used);
    uint32_t pool_at = pool0;
    for (uint32_t k2 = 0; k2 + 1 < nacc; ++k2) {
        const uint32_t l2 = __builtin_amdgcn_readlane(Lk, k2);
        if ((uint32_t)lane > k2) pool_at += (l2 + 3u) & ~3u;
    }
    if ((uint32_t)lane < nacc && (ntok0 + (uint32_t)lane >= YB_MAX_TOKENS || (unsigned long long)pool_at + Lk + 4ull > P.tt.pool_cap)) bad = 1u;
    const unsigned long long badm = __ballot(bad != 0u);
    const uint32_t keep = badm ? (uint32_t)(__ffsll((long long)badm) - 1) : nacc; // merges [0, keep) are committed here
    // (sum of the counts of the merges in front of the last one: a few lanes)
    unsigned long long others_sum = 0ull;
    {
        const unsigned long long mycnt = (uint32_t)lane + 1u < keep ? s_win[wk].cnt : 0ull;
        others_sum = wave_sum_u64(mycnt);
    }
    if ((uint32_t)lane < keep) { // commit merge k = lane
        const uint32_t k = (uint32_t)lane;
        const uint32_t cid = ntok0 + k, key = s_win[wk].key;
        P.tt.off[cid] = pool_at;
        P.tt.len[cid] = Lk;
        P.tt.rec[cid] = TokRec{0u, Lk, Hk, s_acc[k].pc, 0ull};
        P.tt.vset[vslot] = yb_vset_entry(cid, Hk);
        const uint32_t ri = it0 - P.rec_base + k;
        P.rec_left[ri] = key >> 16; // merges.append(best_pair) (trainer.py:296)
        P.rec_right[ri] = key & 0xffffu;
        P.rec_merged[ri] = cid;
        P.rec_count[ri] = s_win[wk].cnt;
        if (P.flat_single && k + 1u < keep) P.rec_sites[ri] = s_win[wk].cnt; // flat layout, a != b: a merge's sites are exactly its count (the last merge of the batch gets what is left, next time)
        P.rec_live_slots[ri] = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(d_live_slots >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)d_live_slots);
        // after the merge no (x, y) adjacency is left anywhere (trainer.py:276-285), so its count is exactly 0: set it here once
        // instead of letting every workgroup subtract its share from one hot address
        P.table.cnt[s_win[wk].slot] = 0ull;
        st->batch[k] = BatchMerge{key >> 16, key & 0xffffu, cid, 1u};
        if (k == 0u) {
            st->a = key >> 16;
            st->b = key & 0xffffu;
            st->c = cid;
            st->c_is_new = 1u;
        }
        if (k + 1u == keep) {
            st->pool_used = (pool_at + Lk + 3u) & ~3u;
            st->n_tokens = cid + 1u;
            st->iter = it0 + keep;
            st->n_batch = keep;
            st->n_select = st->n_select + 1u;
            st->batch_others = others_sum;
            st->best_count = s_win[wk].cnt; // (the host's heuristics look at the lowest count selected so far)
#ifdef YB_PROFILE_LAUNCH
            g_launch_prof[(it0 & 0xFFFFu) * 4 + 3] = wall_clock64(); // (same index as the launch that ran this selection: its st->iter at start)
#endif
        }
    }
    YB_SEL_STAMP(7);
#ifdef YB_PROFILE_SCAN
    if (threadIdx.x == 0 && kmax > 1u && keep != 0u) { // (batch selections only: where the time of a selection goes, by phase of the job)
        unsigned long long *acc = g_sel_acc + (it0 < 3000u ? 0 : it0 < 12000u ? 24 : 48);
        acc[0] += 1ull;
        const int ids[10] = {1, 9, 10, 12, 13, 11, 14, 3, 5, 7};
        for (int q = 0; q < 10; ++q) acc[ids[q]] += g_sel_prof[ids[q]] - g_sel_prof[8];
        acc[16] += keep;
        acc[17] += n_win;
        acc[18] += g_sel_prof[8] - g_sel_prof[0];
    }
#endif
    if (keep != 0u) return;
    // ---- the first merge could not be created blindly: its bytes may be a token already (trainer.py:298-300: then it keeps
    // that id), or there is no id / no pool space left.  Settled here by the whole wave, one merge, the batch ends with it.
    {
        const uint32_t w = s_acc[0].win;
        const uint32_t key = s_win[w].key, x = key >> 16, y = key & 0xffffu, lx = s_win[w].lx, Lm = lx + s_win[w].ly;
        const unsigned long long Hm = yb_hash_concat(s_win[w].hx, s_win[w].hy, s_win[w].ly);
        uint32_t slot = yb_vset_home(Hm, Lm) & P.tt.vset_mask, found = EMPTY;
        while (true) { // (every lane walks the same slots)
            const unsigned long long en = ld_coherent(&P.tt.vset[slot]);
            if (en == VSET_EMPTY) break;
            if ((en >> 32) == (Hm >> 32)) {
                const uint32_t cand = (uint32_t)en;
                const TokRec rc = P.tt.rec[cand];
                if (rc.len == Lm && rc.hash == Hm) {
                    const uint32_t oc = P.tt.off[cand], ox = P.tt.off[x], oy = P.tt.off[y];
                    bool ne = false;
                    for (uint32_t i = (uint32_t)lane; i < Lm; i += 64u)
                        ne |= pool_byte_coherent(P.tt.pool, oc + i) != pool_byte_coherent(P.tt.pool, i < lx ? ox + i : oy + (i - lx));
                    if (!__any(ne)) {
                        found = cand;
                        break;
                    }
                }
            }
            slot = (slot + 1) & P.tt.vset_mask;
        }
        if (lane == 0) {
            uint32_t cid = 0, is_new = 0;
            bool ok = true;
            if (found != EMPTY) {
                cid = found; // bytes already a token: no id is consumed
            } else if (ntok0 >= YB_MAX_TOKENS) {
                st->halt = HALT_VOCAB_FULL;
                ok = false;
            } else if ((unsigned long long)pool0 + Lm + 4ull > P.tt.pool_cap) {
                st->halt = HALT_POOL_FULL;
                ok = false;
            } else {
                cid = ntok0;
                P.tt.off[cid] = pool0;
                P.tt.len[cid] = Lm;
                P.tt.rec[cid] = TokRec{0u, Lm, Hm, s_acc[0].pc, 0ull};
                P.tt.vset[slot] = yb_vset_entry(cid, Hm);
                is_new = 1;
            }
            if (!ok) {
                st->n_batch = 0u;
            } else {
                const uint32_t ri = it0 - P.rec_base;
                P.rec_left[ri] = x;
                P.rec_right[ri] = y;
                P.rec_merged[ri] = cid;
                P.rec_count[ri] = s_win[w].cnt;
                P.rec_live_slots[ri] = d_live_slots;
                P.table.cnt[s_win[w].slot] = 0ull;
                st->a = x;
                st->b = y;
                st->c = cid;
                st->c_is_new = is_new;
                st->best_count = s_win[w].cnt;
                st->batch[0] = BatchMerge{x, y, cid, is_new};
                st->n_batch = 1u;
                st->n_select = st->n_select + 1u;
                st->batch_others = 0ull;
                st->iter = it0 + 1u;
                st->pool_used = is_new ? ((pool0 + Lm + 3u) & ~3u) : pool0;
                st->n_tokens = ntok0 + is_new;
#ifdef YB_PROFILE_LAUNCH
                g_launch_prof[(it0 & 0xFFFFu) * 4 + 3] = wall_clock64();
#endif
            }
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_select(SelectParams P) { select_body(P); }

// ================================================================ loading words into tiles
struct LoadParams {
    const uint8_t *bytes;
    const unsigned long long *off;
    unsigned long long off_base; // off[0]: offsets may be a slice of a larger corpus (word sharding)
    unsigned long long n_words;
    uint16_t *tiles;
    uint32_t *tile_len;
    uint32_t *tile_wbase; // weighted only (else NULL)
    // long words
    uint32_t *long_count; // atomic counter
    unsigned long long *long_total; // atomic: total tokens of long words
    uint32_t *long_word;  // indices of long words (capacity long_cap)
    uint32_t long_cap;
};

__global__ __launch_bounds__(BLOCK) void k_load_words(LoadParams P) {
    const unsigned long long w = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (w >= P.n_words) return;
    const unsigned long long o0 = P.off[w], o1 = P.off[w + 1];
    const unsigned long long L = o1 - o0;
    const unsigned long long pos = (o0 - P.off_base) + w; // packed position: every earlier word contributed its bytes + 1 SEP
    const unsigned long long tile = pos / SPAN;
    const uint32_t slot = (uint32_t)(pos - tile * SPAN);
    uint16_t *dst = P.tiles + tile * CAP + slot;
    const bool weighted = P.tile_wbase != nullptr;
    if (weighted) atomicMin(&P.tile_wbase[tile], (uint32_t)w);
    if (!weighted && L <= 1) return; // flat layout: a word of < 2 tokens can never form a pair (trainer.py:231)
    if (L + 1 > (unsigned long long)LMAX) {
        uint32_t idx = atomicAdd(P.long_count, 1u);
        if (idx < P.long_cap) P.long_word[idx] = (uint32_t)w;
        atomicAdd(P.long_total, L);
        if (weighted) {
            dst[0] = YB_SEP; // placeholder keeps word indices aligned
            atomicMax(&P.tile_len[tile], slot + 1);
        }
        return;
    }
    const uint8_t *src = P.bytes + o0;
    for (uint32_t j = 0; j < (uint32_t)L; ++j) dst[j] = src[j];
    dst[L] = YB_SEP;
    atomicMax(&P.tile_len[tile], slot + (uint32_t)L + 1);
}

struct LoadLongParams {
    const uint8_t *bytes;
    const unsigned long long *off;
    const unsigned long long *wfreq64; // may be NULL
    const uint32_t *long_word;
    const unsigned long long *long_off; // exclusive scan of lengths (host computed)
    uint16_t *long_tok;
    uint32_t *long_len;
    uint32_t *long_freq; // may be NULL
    uint32_t n_long;
};

__global__ __launch_bounds__(BLOCK) void k_load_long(LoadLongParams P) {
    const uint32_t i = blockIdx.x;
    if (i >= P.n_long) return;
    const uint32_t w = P.long_word[i];
    const unsigned long long o0 = P.off[w], L = P.off[w + 1] - o0;
    uint16_t *dst = P.long_tok + P.long_off[i];
    for (unsigned long long j = threadIdx.x; j < L; j += BLOCK) dst[j] = P.bytes[o0 + j];
    if (threadIdx.x == 0) {
        P.long_len[i] = (uint32_t)L;
        if (P.long_freq) P.long_freq[i] = (uint32_t)P.wfreq64[w];
    }
}

__global__ void k_fill_u16(uint16_t *p, unsigned long long n, uint16_t v) {
    // 16 B per thread where possible
    const unsigned long long i = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        uint32_t vv = ((uint32_t)v << 16) | v;
        *reinterpret_cast<uint4 *>(p + i) = make_uint4(vv, vv, vv, vv);
    } else {
        for (unsigned long long j = i; j < n; ++j) p[j] = v;
    }
}

__global__ void k_fill_u32(uint32_t *p, unsigned long long n, uint32_t v) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_freq64_to_32(const unsigned long long *in, uint32_t *out, unsigned long long n, uint32_t *overflow) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long v = in[i];
    if (v > 0xFFFFFFFFull) atomicExch(overflow, 1u);
    out[i] = (uint32_t)v;
}

// sum of tile_len (live slots) -- one block per 4096 tiles is plenty
__global__ __launch_bounds__(BLOCK) void k_sum_u32(const uint32_t *p, unsigned long long n, unsigned long long *out) {
    unsigned long long s = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * BLOCK) s += p[i];
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// ================================================================ candidate argmax (exact, without reading the whole table)
// Invariant kept between two host rebuilds: every slot whose count is >= T is on the candidate list.
//   - k_cand_rebuild (host, now and then): list = all slots with count >= T, T a little below the current best count;
//   - counts only go UP through gt_bump, which appends a slot when its count reaches T (cand_note).
// So the maximum over the list is the true maximum -- with all its ties -- whenever that maximum is >= T.  If it is
// not (the best count decayed below T, or the list overflowed), the selection stops with HALT_RESCAN and the host
// redoes that merge with the full scan.  The best count is non-increasing over merges.
struct CandParams {
    PairTable table;
    const TokRec *rec;
    Best *partials;
    DevState *st;
    CandState *cs;
    uint32_t *ticket;  // != NULL: the last workgroup to finish runs the selection itself (no k_select launch)
    SelectParams sel;
};

// True in exactly one workgroup of the launch: the one that finishes last.  Every workgroup of the grid calls it once,
// when everything it wrote has been acknowledged.  The counter is sharded 8 ways (workgroup b and b + 8 usually share an
// XCD) with a second level of 8 arrivals: several hundred workgroups finishing within a few microseconds would otherwise
// queue on one address (~12 ns per atomic).  Counters reset themselves.
constexpr int TICKET_STRIDE = 32; // u32 words: one 128-B line per counter
constexpr int TICKET_WORDS = 9 * TICKET_STRIDE;
__device__ __forceinline__ bool last_workgroup(uint32_t *ticket) {
    __shared__ uint32_t s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t shard = blockIdx.x & 7u, n_shards = min(8u, gridDim.x);
        const uint32_t mine = (gridDim.x - shard + 7u) >> 3; // workgroups of this shard
        uint32_t last = 0;
        if (__hip_atomic_fetch_add(&ticket[shard * TICKET_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine - 1u) {
            st_coherent(&ticket[shard * TICKET_STRIDE], 0u);
            if (__hip_atomic_fetch_add(&ticket[8 * TICKET_STRIDE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_shards - 1u) {
                st_coherent(&ticket[8 * TICKET_STRIDE], 0u);
                last = 1;
            }
        }
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

// Tail of the fused per-merge launch (k_apply / k_scan_skip with a ticket): the workgroup that finishes last has every
// table update of this merge behind it; it finds the next best pair on the candidate list and runs the selection --
// what k_argmax_cand does as a launch of its own.
struct FuseParams {
    uint32_t *ticket; // NULL: not fused (the selection is a separate launch)
    SelectParams sel;
};
// (call it from ONE place per kernel: the selection is ~9 KB of code, and these launches start with a cold instruction cache)
__device__ __forceinline__ void fused_select_tail(const FuseParams &F, WinEnt *win) {
    if (!F.ticket) return;
    YB_SEL_STAMP(0);
    if (!last_workgroup(F.ticket)) return;
    if (threadIdx.x >= BLOCK) return; // (a wider workgroup: the selection is written for BLOCK threads; ended waves are not waited for)
    YB_SEL_STAMP(8);
    select_eval(F.sel, win);
}

__global__ __launch_bounds__(BLOCK) void k_argmax_cand(CandParams P) {
    __shared__ Best s_b[WPB];
#ifdef YB_PROFILE_SCAN
    if (blockIdx.x == 0) YB_SEL_STAMP(0);
    if (blockIdx.x == 0) YB_SEL_STAMP(8);
#endif
    const uint32_t stop = P.st->done | P.st->halt; // (these loads do not depend on each other: one round trip)
    if (!stop) { // (the list is complete: nothing updates the table while this kernel runs)
        Best best = cand_list_best(P.table, P.rec, min(P.cs->n, CAND_CAP), blockIdx.x * BLOCK + threadIdx.x, gridDim.x * BLOCK).b;
        best = best_wave_reduce(best);
        const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
        if (lane == 0) s_b[wib] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < WPB; ++i)
                if (best_gt(s_b[i], best)) best = s_b[i];
            best_store_coherent(&P.partials[blockIdx.x], best);
        }
#ifdef YB_PROFILE_SCAN
        if (blockIdx.x == 0) YB_SEL_STAMP(9);
#endif
    }
    if (!P.ticket) return;
    // The workgroup that finishes last does the selection.  No __threadfence (an L2 write-back on this part): what
    // the selection reads from THIS kernel -- the partials -- goes through device-scope accesses; each workgroup waits
    // for its own to be acknowledged before it takes its ticket.
    if (!last_workgroup(P.ticket)) return;
    select_body(P.sel);
}

// list = every slot with count >= cs->T (the bitmap was cleared by the host)
__global__ __launch_bounds__(BLOCK) void k_cand_rebuild(CandParams P) {
    const unsigned long long T = P.cs->T;
    for (uint32_t s = blockIdx.x * BLOCK + threadIdx.x; s < P.table.cap; s += gridDim.x * BLOCK) {
        const long long cn = (long long)P.table.cnt[s];
        if (cn <= 0 || (unsigned long long)cn < T) continue;
        const uint32_t idx = atomicAdd(&P.cs->n, 1u);
        if (idx < CAND_CAP) P.table.cand_list[idx] = (unsigned long long)s | ((unsigned long long)P.table.keys[s] << 32); else P.cs->overflow = 1u;
    }
}

// what a workgroup keeps about the merges of the batch (LDS)
struct BatchLds {
    uint32_t a[KMAX], b[KMAX], c[KMAX];
    uint32_t row[KMAX];            // signature block of the pair
    unsigned long long mask[KMAX]; // its three bits inside the block
};

// ================================================================ the per-merge launches (they end with the fused selection)
// ---------------------------------------------------------------- fused form: scan + rewrite in one kernel
// (used while sites are dense: nearly every tile changes, a second read of the stream would cost more)
// HIST (flat layout, at most HIST_V tokens): the deltas go to the direct-indexed LDS store (Hist) instead of the hashed one.
template <bool WEIGHTED, bool HIST = false>
__global__ __launch_bounds__(BLOCK) void k_apply(ApplyParams P, uint32_t apply_blocks, RankParams R, FuseParams F, LongParams LW, uint32_t long_first) {
    using AggV = typename std::conditional<WEIGHTED, unsigned long long, int>::type;
    static_assert(!(HIST && WEIGHTED), "the direct-indexed store holds 32-bit deltas");
    __shared__ uint32_t s_keys[HIST ? 1 : AGG_N];
    __shared__ AggV s_vals[HIST ? 1 : AGG_N];
    __shared__ int s_hist[HIST ? 4 * HIST_V : 1];
    __shared__ __attribute__((aligned(16))) WaveLds s_w[WPB];
    __shared__ unsigned long long s_cnt[2];

    DevState *st = P.st;
    if (st->done | st->halt) return; // (the same answer in every workgroup: nobody takes a ticket)
    // workgroups [0, apply_blocks) apply the merge; [apply_blocks, long_first) do k_rank_update's work and [long_first, ..) the
    // long words' (k_apply_long) in the same launch
    if (blockIdx.x >= long_first) {
        apply_long_block(LW, blockIdx.x - long_first);
    } else if (blockIdx.x >= apply_blocks) {
        rank_update_block(R, blockIdx.x - apply_blocks);
    } else {
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SlowCtx<AggV> C{P, Agg<AggV>{s_keys, s_vals, HIST ? 0u : (uint32_t)AGG_N - 1u}, st, st->a, st->b, st->c, 0u, 0u, lane,
                    KeyMemo{{EMPTY, EMPTY, EMPTY, EMPTY}, {0u, 0u, 0u, 0u}}, HIST ? s_hist : nullptr};
    C.mk = yb_memkey(C.a, C.b);
    C.self = yb_pairkey(C.a, C.b); // its count was set to 0 by k_select: never updated here
    const uint32_t mk = C.mk;
    if constexpr (HIST) hist_init(Hist{s_hist}); else agg_init(C.agg);
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    WaveLds &W = s_w[wib];
    wave_lds_init(W, lane);
    __syncthreads();

    unsigned long long wave_sites = 0; // wave-uniform
    unsigned long long wave_freed = 0; // slots removed from this wave's tiles
#ifdef YB_PROFILE_SLOW
    unsigned long long t_end = 0;
#endif
    const uint32_t stride = apply_blocks * WPB;
    const uint32_t n_tiles = P.n_tiles;
    // A wave walks tiles w, w+stride, w+2*stride, ...; 64 tile lengths are fetched with one vector load and the
    // next tile's 2 x 16 B per lane are in flight while the current tile is examined.
    for (uint32_t batch = blockIdx.x * WPB + wib; batch < n_tiles; batch += stride * 64u) {
        const unsigned long long my_tile = (unsigned long long)batch + (unsigned long long)lane * stride;
        const uint32_t my_len = my_tile < n_tiles ? P.tile_len[my_tile] : 0u;
        const uint32_t cnt = min(64u, (n_tiles - batch + stride - 1) / stride);
        TileRegs nxt = load_tile(P.tiles, batch, __builtin_amdgcn_readlane(my_len, 0), lane);
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint32_t tile = batch + i * stride;
            const uint32_t len = __builtin_amdgcn_readlane(my_len, i);
            const TileRegs r = nxt;
            if (i + 1 < cnt) nxt = load_tile(P.tiles, tile + stride, __builtin_amdgcn_readlane(my_len, i + 1), lane);
            if (len == 0) continue;
            // ---- fast path: does any adjacent pair of this tile equal (a,b)?
            const uint32_t b0 = __builtin_amdgcn_readfirstlane(r.vb.x);
            const uint32_t na = next_lane(r.va.x, b0);
            const uint32_t nb = next_lane(r.vb.x, PADPAD);
            // (nearly every tile of this phase holds the pair: the per-lane candidate masks are computed once, here, and
            // handed to the rewrite -- a cheaper any-test first would be paid on top of them in almost every tile)
            const uint32_t pmA = match_mask8(r.va, na, mk), pmB = match_mask8(r.vb, nb, mk);
            if (!__any((pmA | pmB) != 0u)) continue;
#ifdef YB_PROFILE_SLOW // [5]: from the end of one rewrite to the start of the next (the wait for the tile, the match)
            if (lane == 0 && t_end) W.prof[5] += __builtin_readcyclecounter() - t_end;
#endif
            {
                TileRegs rr = r;
                uint32_t len_io = len, fc = CAP;
                slow_tile<WEIGHTED, AggV, HIST, true>(C, W, tile, len_io, rr, na, nb, wave_sites, wave_freed, fc, pmA, pmB);
            }
#ifdef YB_PROFILE_SLOW
            t_end = __builtin_readcyclecounter();
#endif
        }
    }
#ifdef YB_PROFILE_SLOW
    if (lane < 8 && W.prof[lane]) atomicAdd(&g_slow_prof[lane], W.prof[lane]);
#endif
    apply_epilogue(P, C.agg, st, s_cnt, wave_sites, wave_freed, lane, HIST ? s_hist : nullptr, C.a, C.b, C.c);
    }
    __shared__ WinEnt s_selwin[WIN];
    fused_select_tail(F, s_selwin);
}

// ---------------------------------------------------------------- sparse form: skip index + rewrite, a BATCH of merges per launch
// A workgroup takes runs of `chunk` consecutive tiles; one thread tests a tile's signature once per merge of the batch
// (8 B per tile and merge: the 64-bit block that holds the pair's bits) and notes which merges may have sites there.  The
// tiles that pass (~1 % per merge late in a job) are read by the workgroup's waves in turn; the merges a tile was noted for
// are applied to it IN BATCH ORDER while it sits in registers (single_site_tile / slow_tile, DEFER forms) and the tile
// is written back once.  The selection only batches merges whose relative order cannot matter for anything but this
// (select_batch): a later merge of the batch never touches the tokens an earlier one consumes or creates a pair with.
#ifdef YB_PROFILE_SCAN
#ifdef YB_PROFILE_LAUNCH // (its own build: these same-address atomics sit in front of the workgroups' loads and distort the phase stamps)
#define YB_LAUNCH_START(it) do { if (threadIdx.x == 0) atomicMin(&g_launch_prof[((it) & 0xFFFFu) * 4 + 0], wall_clock64()); } while (0)
#define YB_LAUNCH_END(it) do { if (threadIdx.x == 0) atomicMax(&g_launch_prof[((it) & 0xFFFFu) * 4 + 1], wall_clock64()); } while (0)
#else
#define YB_LAUNCH_START(it) do { } while (0)
#define YB_LAUNCH_END(it) do { } while (0)
#endif
__device__ unsigned long long g_scan_prof[MAX_LISTS_PROF * 8];
#define YB_SCAN_STAMP(i)                                                                       \
    do {                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < MAX_LISTS_PROF) g_scan_prof[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define YB_SCAN_STAMP(i) do { } while (0)
#define YB_LAUNCH_START(it) do { } while (0)
#define YB_LAUNCH_END(it) do { } while (0)
#endif
struct ScanSkipParams {
    ApplyParams A;                // tiles, lengths, where the deltas go, signatures, counters
    unsigned long long *blk_read; // [scan_blocks] tiles actually read (statistics; plain stores)
    uint32_t scan_blocks;         // workgroups [0, scan_blocks) scan; the rest of the grid runs k_rank_update's work
    uint32_t kt;                  // tiles per thread: a workgroup takes `chunk` <= threads * kt consecutive tiles at a time
    uint32_t chunk;               // ... (a multiple of 64; smaller than the workgroup when there are few tiles: a stream of 8,000 tiles
                                  // -- pooled words -- still spreads over 125 workgroups whose waves ALL rewrite matched tiles)
    RankParams R;                 // (lexrank maintenance is independent of the scan: same launch, no extra boundary)
    FuseParams F;                 // ticket != NULL: the workgroup that finishes last selects the next batch
    LongParams LW;                // the long words ride along too: workgroups [long_first, gridDim.x) (k_apply_long's work)
    uint32_t long_first;
};

template <bool WEIGHTED, int NW>
__device__ __forceinline__ bool scan_skip_block(DevState *st, const ScanSkipParams &Q) { // false: stop flag set, nothing done (grid-uniform)
    using AggV = typename std::conditional<WEIGHTED, unsigned long long, int>::type;
    constexpr int NT = NW * 64;                   // threads of the workgroup (NW waves)
    constexpr int KTM = scan_kt_max(NW);
    __shared__ uint32_t s_n, s_claim;             // entries on the hit list; the next one no wave has taken yet
    __shared__ uint2 s_list[NT * KTM];            // (tile, live length | merges noted << 16)
    __shared__ uint32_t s_keys[AGG_N];
    __shared__ AggV s_vals[AGG_N];
    __shared__ __attribute__((aligned(16))) WaveLds s_w[NW];
    __shared__ unsigned long long s_cnt[2];
    __shared__ BatchLds s_bm;
    const ApplyParams &P = Q.A;
    YB_SCAN_STAMP(0);
    // (the batch and the stop flags are read together: one round trip in front of the signature test)
    const uint32_t st_stop = st->done | st->halt;
    const uint32_t nb_ = min(st->n_batch, (uint32_t)KMAX);
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x < Q.scan_blocks && threadIdx.x < (uint32_t)KMAX) {
        const uint32_t k = threadIdx.x;
        BatchMerge m = BatchMerge{0u, 0u, 0u, 0u};
        if (k < nb_) m = st->batch[k];
        const SigHash H = sig_hash(yb_pairkey(m.a, m.b));
        s_bm.a[k] = m.a;
        s_bm.b[k] = m.b;
        s_bm.c[k] = m.c;
        s_bm.row[k] = H.row;
        s_bm.mask[k] = H.mask;
    }
    // (the LDS tables do not depend on the batch: they are cleared while the loads above are on their way)
    if (blockIdx.x < Q.scan_blocks) {
        agg_init(Agg<AggV>{s_keys, s_vals, P.agg_mask}, NT);
        wave_lds_init(s_w[wib], lane);
        if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
        if (threadIdx.x == 0) {
            s_n = 0;
            s_claim = (uint32_t)NW; // (entries 0 .. NW-1 are the waves' first ones)
        }
    }
    if (st_stop) return false; // (the same answer in every workgroup: nobody takes a ticket)
    YB_SCAN_STAMP(5); // (profile build: the batch has arrived)
#ifdef YB_PROFILE_LAUNCH
    const uint32_t prof_it = st->iter;
    YB_LAUNCH_START(prof_it);
#endif
    if (blockIdx.x >= Q.scan_blocks) {
        if (NW > WPB && threadIdx.x >= BLOCK) return false; // (lexrank maintenance and the long words are written for BLOCK threads: the other waves end here)
        if (blockIdx.x >= Q.long_first)
            apply_long_block(Q.LW, blockIdx.x - Q.long_first);
        else
            rank_update_block(Q.R, blockIdx.x - Q.scan_blocks);
        YB_SCAN_STAMP(7);
        YB_LAUNCH_END(prof_it);
        return true;
    }
    const uint32_t n_blocks = Q.scan_blocks;
    const uint32_t nbatch = __builtin_amdgcn_readfirstlane(nb_);
    SlowCtx<AggV> C{P, Agg<AggV>{s_keys, s_vals, P.agg_mask}, st, 0u, 0u, 0u, 0u, 0u, lane,
                    KeyMemo{{EMPTY, EMPTY, EMPTY, EMPTY}, {0u, 0u, 0u, 0u}}};
    WaveLds &W = s_w[wib];
    __syncthreads();
    YB_SCAN_STAMP(6); // (profile build: LDS tables initialised)
    uint32_t n_read = 0; // (tiles this workgroup read in this launch: fits easily)
    unsigned long long wave_sites = 0, wave_freed = 0;
    const uint32_t kt = Q.kt;
    const uint32_t chunk = Q.chunk;
    const uint32_t n_chunks = (P.n_tiles + chunk - 1) / chunk;
    for (uint32_t ch = blockIdx.x; ch < n_chunks; ch += n_blocks) {
        // all of this thread's lengths are requested first, then the signature words merge by merge
        uint32_t len[KTM], km[KTM];
        const uint32_t t_end = min(P.n_tiles, ch * chunk + chunk);
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)KTM; ++j) {
            const uint32_t t = ch * chunk + j * NT + threadIdx.x;
            len[j] = 0;
            km[j] = 0;
            if (j < kt && t < t_end) len[j] = P.tile_len[t]; // (t_end: the end of this workgroup's run of tiles, or of the stream)
        }
        // (KG merges' words in flight at a time -- 16 loads per thread: ONE trip to memory for a batch of up to KG instead of one per
        // two merges; the registers are free here, the rewrite is what sets the kernel's count)
        constexpr uint32_t KG = KTM >= 4 ? 4u : 8u;
        for (uint32_t k0 = 0; k0 < nbatch; k0 += KG) {
            unsigned long long w[KG][KTM];
#pragma unroll
            for (uint32_t kk = 0; kk < KG; ++kk) {
                const bool on = k0 + kk < nbatch; // uniform
                const unsigned long long *rowp = P.sig + (size_t)s_bm.row[(k0 + kk) & (uint32_t)(KMAX - 1)] * P.sig_stride;
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)KTM; ++j) {
                    const uint32_t t = ch * chunk + j * NT + threadIdx.x;
                    w[kk][j] = (on && j < kt && t < t_end) ? rowp[t] : 0ull;
                }
            }
#pragma unroll
            for (uint32_t kk = 0; kk < KG; ++kk) {
                const unsigned long long mask = s_bm.mask[(k0 + kk) & (uint32_t)(KMAX - 1)];
                const bool on = k0 + kk < nbatch;
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)KTM; ++j)
                    if (on && (w[kk][j] & mask) == mask) km[j] |= 1u << (k0 + kk);
            }
        }
        YB_SCAN_STAMP(1);
#pragma unroll
        for (uint32_t j = 0; j < (uint32_t)KTM; ++j) {
            if (j >= kt) break; // uniform
            const bool mb = km[j] != 0 && len[j] != 0;
            const unsigned long long m = __ballot(mb);
            uint32_t base = 0;
            if (lane == 0 && m) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (mb) s_list[base + bits_below_lane(m)] = make_uint2(ch * chunk + j * NT + threadIdx.x, len[j] | (km[j] << 16));
        }
        __syncthreads();
        const uint32_t n = s_n;
        n_read += n;
        // the candidate tiles, dealt to the waves; the next candidate's data is in flight while one is matched and rewritten
        // (a wave's candidates are the same in all its lanes: the descriptors live in scalar registers)
        auto list_at = [&](uint32_t idx) -> uint2 {
            const uint2 v = s_list[idx];
            return make_uint2(__builtin_amdgcn_readfirstlane(v.x), __builtin_amdgcn_readfirstlane(v.y));
        };
        // (a rewritten tile is written back when the NEXT tile's registers have been taken: the wave has to wait for everything it
        // has in flight before it can use a prefetched tile (vmcnt counts in order), and stores issued just before that wait are
        // waited for in full -- ~900 cycles per tile; issued a tile earlier they are long acknowledged)
        TileRegs pend = TileRegs{make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
        uint32_t pend_tile = 0, pend_len = 0, pend_fc = CAP;
        auto write_back = [&](const TileRegs &t, uint32_t tile_, uint32_t len_, uint32_t fc_) {
            // the 16-B groups from the first changed slot to the end of what is live
            const uint32_t pad_end = (len_ + 7u) & ~7u;
            uint4 *wb = reinterpret_cast<uint4 *>(P.tiles + (size_t)tile_ * CAP);
            const uint32_t gA = (uint32_t)lane * 8u, gB = 512u + (uint32_t)lane * 8u;
            if (gA + 8u > fc_ && gA < pad_end) wb[lane] = t.va;
            if (gB + 8u > fc_ && gB < pad_end) wb[64 + lane] = t.vb;
            if (lane == 0) P.tile_len[tile_] = len_;
        };
        // the next entry nobody has taken (tiles differ a lot in what they cost -- a tile that passed the signature test without
        // holding the pair is dropped after the match: dealt out by turns, the waves of a workgroup finish far apart).  Claimed at
        // the END of a tile's turn, where registers are free.
        auto claim = [&]() -> uint32_t {
            uint32_t nx = 0;
            if (lane == 0) nx = atomicAdd(&s_claim, 1u);
            return __builtin_amdgcn_readfirstlane(nx);
        };
        uint32_t j = wib;                // the entry whose tile is in q0
        uint32_t jn = j < n ? claim() : n; // the one after it
        uint2 it0 = j < n ? list_at(j) : make_uint2(0u, 0u);
        TileRegs q0 = load_tile(P.tiles, it0.x, it0.y & 0xffffu, lane);
        while (j < n) {
            const uint32_t tile = it0.x;
            uint32_t tlen = it0.y & 0xffffu, kmask = it0.y >> 16;
            TileRegs r = q0;
            asm volatile("" : "+v"(r.va.x), "+v"(r.vb.x)); // (the prefetched registers are taken HERE: in front of the stores below)
            if (pend_fc < (uint32_t)CAP) {
                write_back(pend, pend_tile, pend_len, pend_fc);
                pend_fc = CAP;
            }
            asm volatile("" ::: "memory"); // (the stores stay here: their registers are free for the next prefetch)
            j = jn;
            if (j < n) { // (one candidate ahead: a second one in flight cost the flat form its fourth wave per SIMD)
                it0 = list_at(j);
                q0 = load_tile(P.tiles, it0.x, it0.y & 0xffffu, lane);
            }
            const uint32_t len_in = tlen;
            uint32_t first_changed = CAP; // wave-uniform: the first slot of the tile that differs from what HBM holds
            while (kmask) { // the merges this tile was noted for, in batch order
                const int k = __ffs((int)kmask) - 1;
                kmask &= kmask - 1u;
                const uint32_t a = __builtin_amdgcn_readfirstlane(s_bm.a[k]), b = __builtin_amdgcn_readfirstlane(s_bm.b[k]); // (uniform: scalars)
                const uint32_t mk = yb_memkey(a, b);
                const uint32_t b0 = __builtin_amdgcn_readfirstlane(r.vb.x);
                const uint32_t na = next_lane(r.va.x, b0);
                const uint32_t nb = next_lane(r.vb.x, PADPAD);
                // (the per-lane site masks at once: an any-test first would be paid on top of them in the 55-75 % of the candidate tiles
                // that do hold the pair)
                const uint32_t mine = match_mask8(r.va, na, mk) | (match_mask8(r.vb, nb, mk) << 8);
                const unsigned long long holders = __ballot(mine != 0);
                if (!holders) continue;
                C.a = a;
                C.b = b;
                C.c = __builtin_amdgcn_readfirstlane(s_bm.c[k]);
                C.mk = mk;
                C.self = yb_pairkey(a, b);
                const int lane_s = __ffsll((long long)holders) - 1;
                const uint32_t mm_s = __builtin_amdgcn_readlane(mine, lane_s < 0 ? 0 : lane_s);
                if (a != b && __popcll(holders) == 1 && __popc(mm_s) == 1) {
                    single_site_tile<AggV, false, WEIGHTED, true>(C, tile, tlen, r, lane_s, mm_s, wave_sites, wave_freed, first_changed);
#ifdef YB_PROFILE_SCAN // how long until everything this tile stored is acknowledged (what a conservative vmcnt(0) costs)
                    {
                        const unsigned long long t0 = __builtin_readcyclecounter();
                        vm_drain();
                        const unsigned long long t1 = __builtin_readcyclecounter();
                        if (lane == 0 && blockIdx.x == 7) {
                            atomicAdd(&g_ss_prof[5], t1 - t0);
                            atomicAdd(&g_ss_prof[6], 1ull);
                        }
                    }
#endif
                } else {
                    slow_tile<WEIGHTED, AggV, false, true, true>(C, W, tile, tlen, r, na, nb, wave_sites, wave_freed, first_changed, mine & 0xffu, mine >> 8); // (the masks are at hand)
                }
            }
            // to be written back, if anything changed (see above; assigned on every path: the registers of the tile before are dead
            // while a tile is rewritten)
            pend = r;
            pend_tile = tile;
            pend_len = tlen;
            pend_fc = first_changed;
            (void)len_in;
            if (j < n) jn = claim();
        }
        if (pend_fc < (uint32_t)CAP) write_back(pend, pend_tile, pend_len, pend_fc);
        __syncthreads();
        YB_SCAN_STAMP(2);
#ifdef YB_PROFILE_SCAN
        if (threadIdx.x == 0 && blockIdx.x < MAX_LISTS_PROF) g_scan_prof[blockIdx.x * 8 + 3] = ((unsigned long long)nbatch << 32) | n; // (merges of this launch, candidate tiles of this workgroup)
#endif
        if (threadIdx.x == 0) {
            // (the two constants are made here: hoisted out of the loop they sat in registers through all of it -- and were the two
            // the 16-wave form had to spill)
            uint32_t zero, first;
            asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
            asm volatile("v_mov_b32 %0, %1" : "=v"(first) : "n"(NW));
            s_n = zero;
            s_claim = first;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && Q.blk_read) Q.blk_read[blockIdx.x] += n_read;
    YB_SCAN_STAMP(4);
    apply_epilogue<AggV, NT>(P, C.agg, st, s_cnt, wave_sites, wave_freed, lane);
    YB_SCAN_STAMP(7);
    YB_LAUNCH_END(prof_it);
    return true;
}
// st (= Q.A.st) is a kernel argument of its own, the FIRST one: the library is built with kernel-argument preload for two
// dwords (Makefile), so the pointer is in SGPRs when the wave starts and the read of the batch record does not have to wait
// for the argument segment first -- one dependent trip less in front of every launch.
template <bool WEIGHTED, int NW>
__global__ __launch_bounds__(NW * 64, 16 / NW) void k_scan_skip(DevState *st, ScanSkipParams Q) { // (NW = 8 or 16; <= 128 VGPRs, 16 waves per CU)
    __shared__ WinEnt s_selwin[WIN];
    if (!scan_skip_block<WEIGHTED, NW>(st, Q)) return;
    fused_select_tail(Q.F, s_selwin);
}

// ================================================================ table growth: re-insert live entries (count != 0)
struct RehashParams {
    PairTable from, to;
    DevState *st;
};
__global__ __launch_bounds__(BLOCK) void k_rehash(RehashParams P) {
    const uint32_t cap = P.from.cap;
    for (uint32_t s = blockIdx.x * BLOCK + threadIdx.x; s < cap; s += gridDim.x * BLOCK) {
        const uint32_t k = P.from.keys[s];
        if (k == EMPTY) continue;
        const long long v = (long long)P.from.cnt[s];
        if (v != 0) gt_add(P.to, P.st, k, v);
    }
}

// ================================================================ multi-GPU: exchange of aggregated pair-count deltas
// Every rank owns a shard of the words and a replica of the pair table.  Its apply pass leaves its pair-count updates as
// [header | (key, delta) records] in its send buffer (the aggregator flush writes them there: flush_entries' record sink);
// the exchange -- k_xchg_push peer to peer, or one all-gather -- brings every rank's buffer to every rank, and k_delta_apply
// adds all of them into the replica.  Integer sums are order-independent, so all replicas hold identical counts and every
// rank selects the same batch of merges.  A rank that must stop (overflow) says so in its header: all ranks then stop at
// the same iteration.

struct DeltaApplyParams {
    const uint8_t *recv; // n_ranks buffers of stride bytes: DeltaHdr, then cap DeltaRec
    uint32_t n_ranks, cap;
    unsigned long long stride;
    PairTable table;
    DevState *st;
    FuseParams F;        // ticket != NULL: the workgroup that finishes last selects the next merge (one launch fewer per merge)
    uint32_t peer_written; // the buffers were written by OTHER processes / devices (peer-to-peer exchange): read them past the caches --
                           // this device's L2s may still hold the lines of the exchange before last
};
// Every rank's records name the same hot pairs over and over (one record per WORKGROUP of the sender that saw the pair, times
// the number of ranks), and same-address atomics are served one per 12.5 ns: added to the table record by record they would
// queue for tens of microseconds at 8 ranks.  A workgroup therefore sums its DELTA_RPW records in the LDS aggregator first
// and flushes that (the same flush as the apply launches': batched key loads, candidate notes).
constexpr int DELTA_RPW = 4 * BLOCK; // records per workgroup
__global__ __launch_bounds__(BLOCK) void k_delta_apply(DeltaApplyParams P) {
    __shared__ uint32_t s_keys[AGG_N];
    __shared__ unsigned long long s_vals[AGG_N];
    if (P.st->done | P.st->halt) return; // (the same answer in every workgroup: nobody takes a ticket)
    const Agg<unsigned long long> agg{s_keys, s_vals, (uint32_t)AGG_N - 1u};
    agg_init(agg);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < DELTA_RPW / BLOCK; ++q) {
        const unsigned long long idx = (unsigned long long)blockIdx.x * DELTA_RPW + (unsigned long long)q * BLOCK + threadIdx.x;
        const uint32_t r = (uint32_t)(idx / P.cap), j = (uint32_t)(idx % P.cap);
        if (r < P.n_ranks) {
            const DeltaHdr *h = reinterpret_cast<const DeltaHdr *>(P.recv + r * P.stride);
            auto ld = [&](const void *q) -> unsigned long long {
                unsigned long long *u = reinterpret_cast<unsigned long long *>(const_cast<void *>(q));
                return P.peer_written ? __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : *u;
            };
            const unsigned long long n = ld(&h->count), hhalt = ld(&h->halt);
            if (j == 0) { // every rank reads every header: all of them stop at the same merge
                if (hhalt) atomicMax(&P.st->halt_req, (uint32_t)hhalt);
                else if (n > P.cap) atomicMax(&P.st->halt_req, (uint32_t)HALT_DELTA_FULL);
                atomicMax(&P.st->xmax, (uint32_t)min(n, 0xffffffffull)); // (every rank sees every header: the same value everywhere)
            }
            if (j < n && j < P.cap) {
                const DeltaRec *rec = reinterpret_cast<const DeltaRec *>(h + 1);
                const unsigned long long w0 = ld(&rec[j]), w1 = ld(reinterpret_cast<const unsigned long long *>(&rec[j]) + 1);
                agg_add(agg, P.table, P.st, (uint32_t)w0, (long long)w1);
            }
        }
    }
    __syncthreads();
    agg_flush<unsigned long long, BLOCK>(agg, P.table, P.st);
    __shared__ WinEnt s_selwin[WIN];
    fused_select_tail(P.F, s_selwin);
}

// ---------------------------------------------------------------- peer-to-peer exchange (instead of the all-gather)
// Every rank maps its peers' receive buffers (hipIpc handles; on one node the mapping goes over xGMI, two processes on ONE
// device work the same way).  After the apply launch one small kernel PUSHES this rank's [header | records] into its slot of
// every peer's buffer and then raises a flag there (sequence number of the exchange; release at system scope); the workgroup
// that copies into the rank's own slot then waits until every peer's flag for this exchange has arrived in ITS memory.  When
// the kernel has finished, everything the next launch (k_delta_apply) reads is in local memory: no collective, no host.
// A rank pushes what it produced (<= cap records), not the whole buffer.  Two buffer halves alternate: a peer can be at most
// one exchange ahead (its next push needs this rank's push of the exchange in between).
constexpr int XCHG_MAX_RANKS = 64;
struct XchgParams {
    const uint8_t *send;              // this rank's [DeltaHdr | cap DeltaRec]
    uint8_t *peer[XCHG_MAX_RANKS];    // receive areas of all ranks as mapped HERE (peer[rank] = the local one)
    unsigned long long flags_off;     // byte offset of the flag words inside a receive area: flags[2][n_ranks] u64
    unsigned long long half_bytes;    // bytes of one buffer half (n_ranks * stride)
    unsigned long long stride;        // bytes of one rank's slot
    unsigned long long seq;           // this exchange's number (>= 1)
    uint32_t rank, n_ranks, cap;
    unsigned long long timeout_ticks; // 100 MHz ticks the wait may take before the job is stopped (a peer died)
    DevState *st;
};
__global__ __launch_bounds__(BLOCK) void k_xchg_push(XchgParams P) {
    const uint32_t p = blockIdx.x; // destination rank
    if (p >= P.n_ranks) return;
    const DeltaHdr *h = reinterpret_cast<const DeltaHdr *>(P.send);
    const unsigned long long n = min(h->count, (unsigned long long)P.cap);
    const unsigned long long words = 2ull + 2ull * n; // u64 words: header + records
    const unsigned long long half = (P.seq & 1ull) * P.half_bytes;
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(P.peer[p] + half + (unsigned long long)P.rank * P.stride);
    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(P.send);
    for (unsigned long long i = threadIdx.x; i < words; i += BLOCK) dst[i] = src[i];
    __threadfence_system(); // (every thread: its stores are written back before the flag can be seen)
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long *flag = reinterpret_cast<unsigned long long *>(P.peer[p] + P.flags_off) + (P.seq & 1ull) * P.n_ranks + P.rank;
        __hip_atomic_store(flag, P.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (p != P.rank) return;
    // the local workgroup: wait for every peer's flag of this exchange (they land in this rank's memory)
    if (threadIdx.x < P.n_ranks) {
        const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(P.peer[P.rank] + P.flags_off) + (P.seq & 1ull) * P.n_ranks + threadIdx.x;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(const_cast<unsigned long long *>(flag), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < P.seq) {
            __builtin_amdgcn_s_sleep(32);
            // a peer is gone: stop the job instead of hanging the device -- and the exchanges queued behind this one do not wait again
            if (ld_coherent(&P.st->halt_req) == (uint32_t)HALT_COMM) break;
            if (wall_clock64() - t0 > P.timeout_ticks) {
                atomicMax(&P.st->halt_req, (uint32_t)HALT_COMM);
                break;
            }
        }
    }
}

// out[i] = sum over rows r of in[r * n + i]
__global__ void k_sum_rows(const unsigned long long *in, unsigned long long *out, uint32_t n, uint32_t rows) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long s = 0;
    for (uint32_t r = 0; r < rows; ++r) s += in[(size_t)r * n + i];
    out[i] = s;
}

// all entries of a table as records (for the global recount exchange)
struct DumpParams {
    PairTable t;
    DeltaRec *rec;
    unsigned long long *count;
    unsigned long long cap;
};
__global__ __launch_bounds__(BLOCK) void k_table_dump(DumpParams P) {
    const uint32_t slots = P.t.cap;
    for (uint32_t s = blockIdx.x * BLOCK + threadIdx.x; s < slots; s += gridDim.x * BLOCK) {
        const uint32_t k = P.t.keys[s];
        if (k == EMPTY) continue;
        const long long v = (long long)P.t.cnt[s];
        if (v == 0) continue;
        const unsigned long long idx = atomicAdd(P.count, 1ull);
        if (idx < P.cap) P.rec[idx] = DeltaRec{k, 0u, v};
    }
}
struct RecApplyParams {
    const DeltaRec *rec;
    unsigned long long n;
    PairTable table;
    DevState *st;
};
__global__ __launch_bounds__(BLOCK) void k_records_apply(RecApplyParams P) {
    const unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i < P.n) gt_add(P.table, P.st, P.rec[i].key, P.rec[i].delta);
}

// ================================================================ debug: compare two pair tables
struct CmpParams {
    PairTable ta, tb;
    unsigned long long *mismatches;
    unsigned long long *dump; // may be NULL
};

__device__ __forceinline__ long long gt_lookup(const PairTable &t, uint32_t key) {
    uint32_t s = pt_home(t, key);
    for (uint32_t probe = 0; probe < t.cap; ++probe) {
        uint32_t k = t.keys[s];
        if (k == key) return (long long)t.cnt[s];
        if (k == EMPTY) return 0;
        s = pt_next(t, s);
    }
    return 0;
}

__global__ __launch_bounds__(BLOCK) void k_table_compare(CmpParams P) {
    const uint32_t cap = P.ta.cap;
    for (uint32_t s = blockIdx.x * BLOCK + threadIdx.x; s < cap; s += gridDim.x * BLOCK) {
        uint32_t k = P.ta.keys[s];
        if (k == EMPTY) continue;
        const long long va = (long long)P.ta.cnt[s], vb = gt_lookup(P.tb, k);
        if (va != vb) {
            const unsigned long long idx = atomicAdd(P.mismatches, 1ull);
            if (P.dump && idx < 16ull) { // (diagnostics: the first few differing keys with both counts)
                P.dump[3 * idx] = k;
                P.dump[3 * idx + 1] = (unsigned long long)va;
                P.dump[3 * idx + 2] = (unsigned long long)vb;
            }
        }
    }
}

// debug: order-independent checksum of the resident words under the current segmentation
struct ChecksumParams {
    const uint16_t *tiles;
    const uint32_t *tile_len;
    const uint32_t *tile_wbase;
    const uint32_t *wfreq;
    uint32_t n_tiles;
    TokTable tt;
    unsigned long long *sum, *words, *tokens;
};

__global__ __launch_bounds__(BLOCK) void k_stream_checksum(ChecksumParams P) {
    // one thread per tile: sequential walk (debug only)
    const uint32_t tile = blockIdx.x * BLOCK + threadIdx.x;
    if (tile >= P.n_tiles) return;
    const uint16_t *t = P.tiles + (size_t)tile * CAP;
    const uint32_t len = P.tile_len[tile];
    unsigned long long h = 1469598103934665603ull, sum = 0, nw = 0, nt = 0;
    uint32_t widx = P.tile_wbase ? P.tile_wbase[tile] : 0;
    uint32_t in_word = 0;
    for (uint32_t p = 0; p < len; ++p) {
        uint32_t v = t[p];
        if (v == YB_PAD) continue;
        if (v == YB_SEP) {
            if (in_word) {
                unsigned long long f = P.wfreq ? P.wfreq[widx] : 1;
                h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
                sum += h * f;
                nw += f;
                nt += (unsigned long long)in_word * f;
            }
            widx++;
            in_word = 0;
            h = 1469598103934665603ull;
            continue;
        }
        // fold the token's bytes, then a token boundary marker
        const uint8_t *pb = P.tt.pool + P.tt.off[v];
        for (uint32_t i = 0; i < P.tt.len[v]; ++i) { h ^= pb[i]; h *= 1099511628211ull; }
        h ^= 0x1ffull; h *= 1099511628211ull;
        in_word++;
    }
    if (sum) atomicAdd(P.sum, sum);
    if (nw) atomicAdd(P.words, nw);
    if (nt) atomicAdd(P.tokens, nt);
}

} // namespace yb
