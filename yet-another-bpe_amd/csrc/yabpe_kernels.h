// yabpe_kernels.h -- HIP kernels of the BPE training hot path for gfx950 (MI355X, CDNA4).
//
// One translation unit (yabpe.hip) includes this header.  DESIGN.md describes the data layout and each
// kernel's roofline; the short version:
//
//   token stream  u16 ids in tiles of YB_CAP slots (2 KiB); one wave owns one tile at a time; words never
//                 straddle tiles, every word ends with SEP, PAD fills lead/tail.  tile_len[t] = live slots.
//   pair table    open-addressing hash in HBM: keys u32 (left<<16|right), counts i64, updated with
//                 device-scope atomics; per-workgroup LDS hash aggregates deltas first.
//   k_apply       THE hot kernel: one coalesced read of the live token stream per merge (16 B per lane),
//                 match (a,b) on packed dwords, and only in tiles that contain a site: rewrite + in-tile
//                 compaction + pair-count deltas (tile_logic.h).  HBM-bound; no MFMA (integer indexing work).
//   k_argmax_*    max over the table by (count, lexrank[left], lexrank[right])  (trainer.py:246)
//   k_select      stop rules, merged-token creation / byte-string identity (trainer.py:241-251, 296-300)
//   k_rank_update keeps lexrank[] = rank of every token's bytes in Python bytes order
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
constexpr int LONG_CH = 4096;        // long-word path: tokens per LDS chunk

enum : uint32_t { HALT_NONE = 0, HALT_TABLE_FULL = 1, HALT_POOL_FULL = 2, HALT_VOCAB_FULL = 3, HALT_DELTA_FULL = 4 };

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
    uint32_t pad0;
    unsigned long long min_freq;
    unsigned long long table_entries;
    unsigned long long live_slots; // sum of tile_len
    unsigned long long sites;      // sites merged since the last k_select
    unsigned long long tokens_now; // T_i
    unsigned long long delta_entries;
};

struct PairTable {
    uint32_t *keys;
    unsigned long long *cnt;
    uint32_t mask;
    uint32_t max_probe;
    unsigned long long *entries; // where successful inserts are counted
};

struct Best {
    unsigned long long cnt;
    uint32_t rk;  // lexrank[left] << 16 | lexrank[right]
    uint32_t key; // left << 16 | right
};

__device__ __forceinline__ uint32_t hash32(uint32_t k) {
    k ^= k >> 16;
    k *= 0x7feb352dU;
    k ^= k >> 15;
    k *= 0x846ca68bU;
    k ^= k >> 16;
    return k;
}

// Within one wave LDS operations complete in order; this keeps the compiler from moving a lane's LDS reads
// above other lanes' LDS writes.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ unsigned long long lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// ---------------------------------------------------------------- global pair table
__device__ __forceinline__ void gt_add(const PairTable &t, DevState *st, uint32_t key, long long d) {
    uint32_t s = hash32(key) & t.mask;
    for (uint32_t probe = 0; probe < t.max_probe; ++probe) {
        uint32_t k = __hip_atomic_load(&t.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k == EMPTY) {
            k = atomicCAS(&t.keys[s], EMPTY, key);
            if (k == EMPTY) {
                atomicAdd(t.entries, 1ull);
                k = key;
            }
        }
        if (k == key) {
            atomicAdd(&t.cnt[s], (unsigned long long)d);
            return;
        }
        s = (s + 1) & t.mask;
    }
    atomicMax(&st->halt_req, (uint32_t)HALT_TABLE_FULL);
}

// ---------------------------------------------------------------- LDS aggregator (per workgroup)
struct Agg {
    uint32_t *keys;
    unsigned long long *vals;
};

__device__ __forceinline__ void agg_init(Agg g) {
    for (int i = threadIdx.x; i < AGG_N; i += BLOCK) {
        g.keys[i] = EMPTY;
        g.vals[i] = 0ull;
    }
}

__device__ __forceinline__ void agg_add(Agg g, const PairTable &t, DevState *st, uint32_t key, long long d) {
    uint32_t s = hash32(key) & (AGG_N - 1);
#pragma unroll 1
    for (int probe = 0; probe < 8; ++probe) {
        uint32_t k = __hip_atomic_load(&g.keys[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (k == EMPTY) {
            k = atomicCAS(&g.keys[s], EMPTY, key);
            if (k == EMPTY) k = key;
        }
        if (k == key) {
            atomicAdd(&g.vals[s], (unsigned long long)d);
            return;
        }
        s = (s + 1) & (AGG_N - 1);
    }
    gt_add(t, st, key, d); // aggregator full around this hash: go straight to HBM
}

__device__ __forceinline__ void agg_flush(Agg g, const PairTable &t, DevState *st) {
    for (int i = threadIdx.x; i < AGG_N; i += BLOCK) {
        uint32_t k = g.keys[i];
        long long v = (long long)g.vals[i];
        if (k != EMPTY && v != 0) gt_add(t, st, k, v);
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
    bool m = (v.x == mk) | (v.y == mk) | (v.z == mk) | (v.w == mk);
    m |= __builtin_amdgcn_alignbit(v.y, v.x, 16) == mk;
    m |= __builtin_amdgcn_alignbit(v.z, v.y, 16) == mk;
    m |= __builtin_amdgcn_alignbit(v.w, v.z, 16) == mk;
    m |= __builtin_amdgcn_alignbit(nxt, v.w, 16) == mk;
    return m;
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
            unsigned long long m = __ballot((T(p) == a) & (T(p + 1) == b));
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
    Agg agg{s_keys, s_vals};
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
                uint32_t widx = sep_carry + __popcll(sm & lanemask_lt(lane));
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
    PairTable out; // where deltas go: the pair table (1 GPU) or the per-rank delta table (multi-GPU)
    DevState *st;
};

template <bool WEIGHTED>
__global__ __launch_bounds__(BLOCK) void k_apply(ApplyParams P) {
    __shared__ uint32_t s_keys[AGG_N];
    __shared__ unsigned long long s_vals[AGG_N];
    __shared__ __attribute__((aligned(16))) uint16_t s_stage[WPB][8 + CAP + 8];
    __shared__ __attribute__((aligned(16))) uint16_t s_out[WPB][CAP];
    __shared__ unsigned long long s_mb[WPB][CAP / 64 + 2];

    DevState *st = P.st;
    if (st->done | st->halt) return;
    const uint32_t a = st->a, b = st->b, c = st->c;
    const uint32_t mk = yb_memkey(a, b);

    Agg agg{s_keys, s_vals};
    agg_init(agg);
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint16_t *stg = s_stage[wib];
    uint16_t *outb = s_out[wib];
    unsigned long long *mb = s_mb[wib];
    if (lane < 8) {
        stg[lane] = YB_PAD;
        stg[8 + CAP + lane] = YB_PAD;
    }
    __syncthreads();
    TokAt T{stg};
    MrgAt M{mb};

    unsigned long long wave_sites = 0; // wave-uniform
    unsigned long long wave_freed = 0; // slots removed from this wave's tiles
    const uint32_t stride = gridDim.x * WPB;
    for (uint32_t tile = blockIdx.x * WPB + wib; tile < P.n_tiles; tile += stride) {
        const uint32_t len = P.tile_len[tile];
        if (len == 0) continue;
        TileRegs r = load_tile(P.tiles, tile, len, lane);
        // ---- fast path: does any adjacent pair of this tile equal (a,b)?
        uint32_t na = __shfl_down(r.va.x, 1);
        uint32_t nb = __shfl_down(r.vb.x, 1);
        uint32_t b0 = __shfl(r.vb.x, 0);
        if (lane == 63) {
            na = b0;
            nb = PADPAD;
        }
        bool hit = match4(r.va, na, mk) | match4(r.vb, nb, mk);
        if (!__any(hit)) continue;

        // ---- slow path: this tile has at least one candidate site
        wave_sync();
        stage_tile(stg, r, lane);
        wave_sync();
        const int rounds = (len + 63) >> 6;
        unsigned long long any = mark_sites(stg, mb, rounds, a, b, lane);
        wave_sync();
        if (!any) continue; // (a == b only: a lone candidate can lose to the parity rule -- never for a != b)

        uint32_t outpos = 0;
        uint32_t sep_carry = WEIGHTED ? P.tile_wbase[tile] : 0;
        for (int k = 0; k < rounds; ++k) {
            const int p = k * 64 + lane;
            const unsigned long long mword = mb[1 + k];
            uint32_t widx = 0;
            if (WEIGHTED) {
                unsigned long long sm = __ballot(T(p) == YB_SEP);
                widx = sep_carry + __popcll(sm & lanemask_lt(lane));
                sep_carry += __popcll(sm);
            }
            if (mword) {
                if ((mword >> lane) & 1ull) {
                    YbDeltas d;
                    yb_site_deltas(p, a, b, c, T, M, d);
                    long long w = WEIGHTED ? (long long)P.wfreq[widx] : 1;
                    for (int i = 0; i < d.n; ++i) agg_add(agg, P.out, st, d.key[i], d.sign[i] * w);
                    if (WEIGHTED) agg_add(agg, P.out, st, yb_pairkey(a, b), -w);
                }
                wave_sites += __popcll(mword);
            }
            uint32_t o = 0;
            int keep = yb_keep(p, c, !WEIGHTED, T, M, o);
            unsigned long long km = __ballot(keep);
            if (keep) outb[outpos + __popcll(km & lanemask_lt(lane))] = (uint16_t)o;
            outpos += __popcll(km);
        }
        const uint32_t new_len = outpos;
        const uint32_t pad_end = (new_len + 7u) & ~7u;
        if (lane < 8 && new_len + lane < pad_end) outb[new_len + lane] = YB_PAD;
        wave_sync();
        uint4 *wb = reinterpret_cast<uint4 *>(P.tiles + (size_t)tile * CAP);
        if ((uint32_t)(lane * 8) < new_len) wb[lane] = *reinterpret_cast<const uint4 *>(outb + lane * 8);
        if ((uint32_t)(512 + lane * 8) < new_len) wb[64 + lane] = *reinterpret_cast<const uint4 *>(outb + 512 + lane * 8);
        if (lane == 0) P.tile_len[tile] = new_len;
        wave_freed += len - new_len;
    }
    if (lane == 0) {
        if (wave_sites) {
            if (!WEIGHTED) agg_add(agg, P.out, st, yb_pairkey(a, b), -(long long)wave_sites);
            atomicAdd(&st->sites, wave_sites);
        }
        if (wave_freed) atomicAdd(&st->live_slots, (unsigned long long)(-(long long)wave_freed));
    }
    __syncthreads();
    agg_flush(agg, P.out, st);
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
};

__global__ __launch_bounds__(BLOCK) void k_count_long(LongParams P) {
    const uint32_t i = blockIdx.x;
    if (i >= P.n_long) return;
    const uint16_t *t = P.tok + P.off[i];
    const uint32_t len = P.len[i];
    const long long w = P.freq ? (long long)P.freq[i] : 1;
    for (uint32_t p = threadIdx.x; p + 1 < len; p += BLOCK) gt_add(P.out, P.st, yb_pairkey(t[p], t[p + 1]), w);
}

__global__ __launch_bounds__(BLOCK) void k_apply_long(LongParams P) {
    __shared__ uint16_t s_in[LONG_CH + 4];
    __shared__ uint16_t s_o[LONG_CH];
    __shared__ uint32_t s_j, s_o_pos, s_adv, s_nout;
    DevState *st = P.st;
    if (st->done | st->halt) return;
    const uint32_t i = blockIdx.x;
    if (i >= P.n_long) return;
    const uint32_t a = st->a, b = st->b, c = st->c;
    uint16_t *t = P.tok + P.off[i];
    const uint32_t len = P.len[i];
    const long long w = P.freq ? (long long)P.freq[i] : 1;
    int any = 0;
    for (uint32_t p = threadIdx.x; p + 1 < len; p += BLOCK) any |= (t[p] == a) & (t[p + 1] == b);
    if (!__syncthreads_or(any)) return;

    if (threadIdx.x == 0) {
        s_j = 0;
        s_o_pos = 0;
    }
    // thread 0 carries the sequential state of the greedy rewrite across chunks
    bool have_prev = false;
    uint32_t prev_old = 0, prev_new = 0;
    unsigned long long sites = 0;
    __syncthreads();
    while (true) {
        const uint32_t j = s_j;
        if (j >= len) break;
        const uint32_t n_in = min((uint32_t)(LONG_CH + 3), len - j);
        for (uint32_t q = threadIdx.x; q < n_in; q += BLOCK) s_in[q] = t[j + q];
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t q = 0, no = 0;
            while (q < (uint32_t)LONG_CH && j + q < len) {
                if (j + q + 1 < len && s_in[q] == a && s_in[q + 1] == b) {
                    if (have_prev) {
                        gt_add(P.out, st, yb_pairkey(prev_old, a), -w);
                        gt_add(P.out, st, yb_pairkey(prev_new, c), +w);
                    }
                    gt_add(P.out, st, yb_pairkey(a, b), -w);
                    if (j + q + 2 < len) {
                        bool next_site = (j + q + 3 < len) && s_in[q + 2] == a && s_in[q + 3] == b;
                        if (!next_site) {
                            gt_add(P.out, st, yb_pairkey(b, s_in[q + 2]), -w);
                            gt_add(P.out, st, yb_pairkey(c, s_in[q + 2]), +w);
                        }
                    }
                    s_o[no++] = (uint16_t)c;
                    prev_old = b;
                    prev_new = c;
                    have_prev = true;
                    q += 2;
                    sites++;
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
        if (sites) atomicAdd(&st->sites, sites);
    }
}

// ================================================================ argmax (trainer.py:246)
__device__ __forceinline__ bool best_gt(const Best &x, const Best &y) {
    return x.cnt > y.cnt || (x.cnt == y.cnt && x.rk > y.rk);
}

__device__ __forceinline__ Best best_wave_reduce(Best v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        Best u;
        u.cnt = __shfl_xor(v.cnt, o);
        u.rk = __shfl_xor(v.rk, o);
        u.key = __shfl_xor(v.key, o);
        if (best_gt(u, v)) v = u;
    }
    return v;
}

struct ArgmaxParams {
    PairTable table;
    const uint32_t *rank;
    Best *partials; // one per block
    DevState *st;
};

__global__ __launch_bounds__(BLOCK) void k_argmax_partial(ArgmaxParams P) {
    __shared__ Best s_b[WPB];
    if (P.st->done | P.st->halt) return;
    Best best{0ull, 0u, EMPTY};
    const uint32_t cap = P.table.mask + 1;
    for (uint32_t s = blockIdx.x * BLOCK + threadIdx.x; s < cap; s += gridDim.x * BLOCK) {
        uint32_t k = P.table.keys[s];
        if (k == EMPTY) continue;
        long long cn = (long long)P.table.cnt[s];
        if (cn <= 0 || (unsigned long long)cn < best.cnt) continue;
        Best e{(unsigned long long)cn, (P.rank[k >> 16] << 16) | P.rank[k & 0xffffu], k};
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

// ---------------------------------------------------------------- token byte strings on the device
struct TokTable {
    uint8_t *pool;
    uint32_t *off;
    uint32_t *len;
    uint32_t *rank;  // lexrank[id]
    uint32_t *vset;  // open-addressing set of ids keyed by the token bytes
    uint32_t vset_mask;
    uint32_t pool_cap;
};

// Hash of a byte string from its length and its first/last <= 16 bytes (cheap for very long tokens;
// equality is always decided by a full compare).  Must be identical on host and device.
YB_HD uint32_t yb_tok_hash(const uint8_t *p, uint32_t n) {
    uint32_t h = 2166136261u ^ n;
    uint32_t m = n < 16u ? n : 16u;
    for (uint32_t i = 0; i < m; ++i) h = (h ^ p[i]) * 16777619u;
    for (uint32_t i = 0; i < m; ++i) h = (h ^ p[n - 1 - i]) * 16777619u;
    h ^= h >> 15;
    h *= 0x2c1b3c6dU;
    h ^= h >> 12;
    return h;
}

// Python bytes order: unsigned bytewise, a proper prefix sorts lower.
__device__ __forceinline__ int tok_cmp(const TokTable &tt, uint32_t x, uint32_t y) {
    const uint8_t *px = tt.pool + tt.off[x], *py = tt.pool + tt.off[y];
    const uint32_t lx = tt.len[x], ly = tt.len[y];
    const uint32_t n = lx < ly ? lx : ly;
    for (uint32_t i = 0; i < n; ++i) {
        int d = (int)px[i] - (int)py[i];
        if (d) return d;
    }
    return (lx > ly) - (lx < ly);
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
};

__global__ __launch_bounds__(BLOCK) void k_select(SelectParams P) {
    __shared__ Best s_b[BLOCK];
    __shared__ uint32_t s_flag, s_x, s_y, s_lx, s_L, s_pu, s_slot, s_cand, s_eq;
    DevState *st = P.st;
    const int tid = threadIdx.x;
    if (tid == 0) {
        if (st->halt == 0 && st->halt_req != 0) st->halt = st->halt_req;
        s_flag = st->done | st->halt;
    }
    __syncthreads();
    if (s_flag) return;
    Best best{0ull, 0u, EMPTY};
    for (uint32_t i = tid; i < P.n_partials; i += BLOCK)
        if (best_gt(P.partials[i], best)) best = P.partials[i];
    s_b[tid] = best;
    __syncthreads();
    for (int o = BLOCK / 2; o >= 1; o >>= 1) {
        if (tid < o && best_gt(s_b[tid + o], s_b[tid])) s_b[tid] = s_b[tid + o];
        __syncthreads();
    }
    if (tid == 0) {
        best = s_b[0];
        // close the log entry of the previous iteration
        const uint32_t it = st->iter;
        if (it > P.rec_base) P.rec_sites[it - 1 - P.rec_base] = st->sites;
        st->tokens_now -= st->sites;
        st->sites = 0;
        // stop rules: iteration limit (trainer.py:241), no pairs (:242-243), min_frequency (:247-248)
        if (it >= st->num_merges || best.cnt == 0 || best.cnt < st->min_freq) {
            st->done = 1;
            s_flag = 1;
        } else {
            const uint32_t x = best.key >> 16, y = best.key & 0xffffu;
            const uint32_t lx = P.tt.len[x], ly = P.tt.len[y];
            if ((unsigned long long)st->pool_used + lx + ly > P.tt.pool_cap) {
                st->halt = HALT_POOL_FULL;
                s_flag = 1;
            }
            s_x = x;
            s_y = y;
            s_lx = lx;
            s_L = lx + ly;
            s_pu = st->pool_used;
        }
    }
    __syncthreads();
    if (s_flag) return;
    const uint32_t x = s_x, y = s_y, lx = s_lx, L = s_L, pu = s_pu;
    // merged = p0 + p1 (trainer.py:251), written at the end of the pool
    uint8_t *mp = P.tt.pool + pu;
    {
        const uint8_t *px = P.tt.pool + P.tt.off[x], *py = P.tt.pool + P.tt.off[y];
        for (uint32_t i = tid; i < L; i += BLOCK) mp[i] = i < lx ? px[i] : py[i - lx];
    }
    __threadfence_block();
    __syncthreads();
    // "merged not in vocab" (trainer.py:298): probe the byte-string set
    if (tid == 0) s_slot = yb_tok_hash(mp, L) & P.tt.vset_mask;
    __syncthreads();
    uint32_t found = EMPTY;
    while (true) {
        if (tid == 0) {
            s_cand = P.tt.vset[s_slot];
            s_eq = 1;
        }
        __syncthreads();
        const uint32_t cand = s_cand;
        if (cand == EMPTY) break;
        if (P.tt.len[cand] == L) {
            const uint8_t *pc = P.tt.pool + P.tt.off[cand];
            int ne = 0;
            for (uint32_t i = tid; i < L; i += BLOCK) ne |= pc[i] != mp[i];
            if (ne) s_eq = 0; // benign race: every writer stores 0
            __syncthreads();
            if (s_eq) {
                found = cand;
                break;
            }
        }
        __syncthreads();
        if (tid == 0) s_slot = (s_slot + 1) & P.tt.vset_mask;
        __syncthreads();
    }
    if (tid == 0) {
        uint32_t cid;
        uint32_t is_new = 0;
        if (found != EMPTY) {
            cid = found; // bytes already a token: no id is consumed (trainer.py:298-300)
        } else if (st->n_tokens >= YB_MAX_TOKENS) {
            st->halt = HALT_VOCAB_FULL;
            return;
        } else {
            cid = st->n_tokens;
            P.tt.off[cid] = pu;
            P.tt.len[cid] = L;
            P.tt.rank[cid] = 0;
            P.tt.vset[s_slot] = cid;
            st->pool_used = pu + L;
            st->n_tokens = cid + 1;
            is_new = 1;
        }
        const uint32_t it = st->iter;
        const uint32_t ri = it - P.rec_base;
        P.rec_left[ri] = x; // merges.append(best_pair) (trainer.py:296)
        P.rec_right[ri] = y;
        P.rec_merged[ri] = cid;
        P.rec_count[ri] = s_b[0].cnt;
        P.rec_live_slots[ri] = st->live_slots;
        st->a = x;
        st->b = y;
        st->c = cid;
        st->c_is_new = is_new;
        st->iter = it + 1;
    }
}

// lexrank maintenance after a new token c was created: tokens above it move up by one, and c's rank is the
// number of tokens below it.
struct RankParams {
    TokTable tt;
    DevState *st;
};

__global__ __launch_bounds__(BLOCK) void k_rank_update(RankParams P) {
    __shared__ uint32_t s_less;
    DevState *st = P.st;
    if (st->done | st->halt) return;
    if (!st->c_is_new) return;
    const uint32_t n = st->n_tokens, c = st->c;
    if (blockIdx.x * BLOCK >= n) return;
    if (threadIdx.x == 0) s_less = 0;
    __syncthreads();
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    int less = 0;
    if (t < n && t != c) {
        int cmp = tok_cmp(P.tt, t, c);
        if (cmp > 0)
            P.tt.rank[t] += 1;
        else
            less = 1;
    }
    unsigned long long m = __ballot(less);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&s_less, (uint32_t)__popcll(m));
    __syncthreads();
    if (threadIdx.x == 0 && s_less) atomicAdd(&P.tt.rank[c], s_less);
}

// ================================================================ loading words into tiles
struct LoadParams {
    const uint8_t *bytes;
    const unsigned long long *off;
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
    const unsigned long long pos = o0 + w; // packed position: every earlier word contributed its bytes + 1 SEP
    const unsigned long long tile = pos / SPAN;
    const uint32_t slot = (uint32_t)(pos - tile * SPAN);
    uint16_t *dst = P.tiles + tile * CAP + slot;
    if (P.tile_wbase) atomicMin(&P.tile_wbase[tile], (uint32_t)w);
    if (L + 1 > (unsigned long long)LMAX) {
        uint32_t idx = atomicAdd(P.long_count, 1u);
        if (idx < P.long_cap) P.long_word[idx] = (uint32_t)w;
        atomicAdd(P.long_total, L);
        dst[0] = YB_SEP; // placeholder keeps word indices aligned
        atomicMax(&P.tile_len[tile], slot + 1);
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

// ================================================================ debug: compare two pair tables
struct CmpParams {
    PairTable ta, tb;
    unsigned long long *mismatches;
};

__device__ __forceinline__ long long gt_lookup(const PairTable &t, uint32_t key) {
    uint32_t s = hash32(key) & t.mask;
    for (uint32_t probe = 0; probe <= t.mask; ++probe) {
        uint32_t k = t.keys[s];
        if (k == key) return (long long)t.cnt[s];
        if (k == EMPTY) return 0;
        s = (s + 1) & t.mask;
    }
    return 0;
}

__global__ __launch_bounds__(BLOCK) void k_table_compare(CmpParams P) {
    const uint32_t cap = P.ta.mask + 1;
    for (uint32_t s = blockIdx.x * BLOCK + threadIdx.x; s < cap; s += gridDim.x * BLOCK) {
        uint32_t k = P.ta.keys[s];
        if (k == EMPTY) continue;
        if ((long long)P.ta.cnt[s] != gt_lookup(P.tb, k)) atomicAdd(P.mismatches, 1ull);
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
