// yabpe_aux_kernels.h -- one-off passes around the hot loop: prefix sums, the synthetic corpus generator
// (SURVEY.md 8d), device-side pooling of equal words (trainer.py:221-225) and retiling of the shrinking
// token stream.  None of these is on the per-merge critical path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "yabpe_kernels.h"

namespace yb {

#define YB_RET(call)                         \
    do {                                     \
        if ((call) != hipSuccess) return -1; \
    } while (0)

// ================================================================ exclusive prefix sum (u64 out)
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

// out[i] = sum_{k<i} in[k] for i in [0, n_out); in[k] = 0 for k >= n_in.  block_sums[b] = sum of block b.
template <class TIn>
__global__ __launch_bounds__(BLOCK) void k_scan_block(const TIn *in, unsigned long long n_in, unsigned long long *out,
                                                      unsigned long long n_out, unsigned long long *block_sums) {
    __shared__ unsigned long long s_w[WPB];
    const unsigned long long base = (unsigned long long)blockIdx.x * SCAN_TILE + (unsigned long long)threadIdx.x * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS];
    unsigned long long tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        unsigned long long i = base + k;
        v[k] = i < n_in ? (unsigned long long)in[i] : 0ull;
        tsum += v[k];
    }
    // wave inclusive scan of thread sums
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    unsigned long long inc = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        unsigned long long u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    if (lane == 63) s_w[wib] = inc;
    __syncthreads();
    unsigned long long woff = 0;
    for (int w = 0; w < wib; ++w) woff += s_w[w];
    unsigned long long run = woff + inc - tsum;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        unsigned long long i = base + k;
        if (i < n_out) out[i] = run;
        run += v[k];
    }
    if (threadIdx.x == BLOCK - 1 && block_sums) block_sums[blockIdx.x] = woff + inc;
}

__global__ __launch_bounds__(BLOCK) void k_scan_add(unsigned long long *out, unsigned long long n_out,
                                                    const unsigned long long *block_off) {
    const unsigned long long add = block_off[blockIdx.x];
    const unsigned long long base = (unsigned long long)blockIdx.x * SCAN_TILE + (unsigned long long)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n_out) out[base + k] += add;
}

// out must hold n_out >= 1 elements (typically n_in + 1, so that out[n_in] is the total).
template <class TIn>
int exclusive_scan(hipStream_t s, const TIn *in, unsigned long long n_in, unsigned long long *out, unsigned long long n_out) {
    const unsigned long long nb = (n_out + SCAN_TILE - 1) / SCAN_TILE;
    if (nb <= 1) {
        hipLaunchKernelGGL(k_scan_block<TIn>, dim3(1), dim3(BLOCK), 0, s, in, n_in, out, n_out, (unsigned long long *)nullptr);
        YB_RET(hipGetLastError());
        return 0;
    }
    unsigned long long *sums = nullptr, *sums_scan = nullptr;
    YB_RET(hipMalloc((void **)&sums, nb * 8));
    YB_RET(hipMalloc((void **)&sums_scan, nb * 8));
    hipLaunchKernelGGL(k_scan_block<TIn>, dim3((uint32_t)nb), dim3(BLOCK), 0, s, in, n_in, out, n_out, sums);
    YB_RET(hipGetLastError());
    int r = exclusive_scan<unsigned long long>(s, sums, nb, sums_scan, nb);
    if (r == 0) {
        hipLaunchKernelGGL(k_scan_add, dim3((uint32_t)nb), dim3(BLOCK), 0, s, out, n_out, sums_scan);
        if (hipGetLastError() != hipSuccess) r = -1;
    }
    if (hipStreamSynchronize(s) != hipSuccess) r = -1;
    (void)hipFree(sums);
    (void)hipFree(sums_scan);
    return r;
}

// ================================================================ synthetic corpus (mirrors yet_another_bpe/synth.py)
__host__ __device__ __forceinline__ unsigned long long synth_mix(unsigned long long x) {
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return x;
}
__host__ __device__ __forceinline__ unsigned long long synth_rnd(unsigned long long seed, unsigned long long stream, unsigned long long i) {
    return synth_mix(seed + 0x9E3779B97F4A7C15ull * (i + 1) + 0xD1B54A32D192ED03ull * stream);
}

__global__ void k_synth_lexicon(unsigned long long seed, uint32_t n_types, const uint8_t *alphabet, uint32_t alen,
                                uint8_t *type_bytes, uint8_t *type_len) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_types) return;
    type_len[j] = (uint8_t)(1 + synth_rnd(seed, 1, j) % 12);
    for (uint32_t k = 0; k < 12; ++k) type_bytes[(size_t)j * 12 + k] = alphabet[synth_rnd(seed, 2, 16ull * j + k) % alen];
}

__global__ void k_synth_draw(unsigned long long seed, unsigned long long n_cand, const unsigned long long *cum, uint32_t n_types,
                             const uint8_t *type_len, uint32_t prefix, uint32_t *word_type, uint8_t *word_len) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand) return;
    const unsigned long long u = synth_rnd(seed, 3, i) % cum[n_types - 1];
    uint32_t lo = 0, hi = n_types;  // first j with cum[j] > u
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (cum[mid] > u) hi = mid; else lo = mid + 1;
    }
    word_type[i] = lo;
    word_len[i] = (uint8_t)(type_len[lo] + prefix);
}

// smallest n >= 1 with off[n] >= target (off is the exclusive scan, off[n] = bytes of the first n words)
__global__ void k_synth_cut(const unsigned long long *off, unsigned long long n_cand, unsigned long long target, unsigned long long *out) {
    unsigned long long lo = 1, hi = n_cand;
    if (off[n_cand] < target) { out[0] = 0; out[1] = off[n_cand]; return; }
    while (lo < hi) {
        unsigned long long mid = (lo + hi) >> 1;
        if (off[mid] >= target) hi = mid; else lo = mid + 1;
    }
    out[0] = lo;
    out[1] = off[lo];
}

__global__ void k_synth_fill(const uint32_t *word_type, const unsigned long long *off, unsigned long long n_words,
                             const uint8_t *type_bytes, const uint8_t *type_len, uint32_t prefix, uint8_t *bytes) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const uint32_t ty = word_type[i];
    uint8_t *dst = bytes + off[i];
    if (prefix) *dst++ = 0x20;
    const uint8_t *src = type_bytes + (size_t)ty * 12;
    const uint32_t n = type_len[ty];
    for (uint32_t k = 0; k < n; ++k) dst[k] = src[k];
}

struct SynthOut {
    uint8_t *bytes;
    unsigned long long *off;
    unsigned long long n_words, n_bytes;
};

inline int synth_generate(hipStream_t s, unsigned long long target, uint32_t n_types, unsigned long long seed,
                          const uint8_t *alphabet, uint32_t alen, int space_prefix, SynthOut *out) {
    const uint32_t prefix = space_prefix ? 1 : 0;
    const unsigned long long n_cand = target / 4 + 4096;
    uint8_t *d_alpha = nullptr, *d_tb = nullptr, *d_tl = nullptr, *d_wl = nullptr, *d_bytes = nullptr;
    unsigned long long *d_cum = nullptr, *d_off = nullptr, *d_cut = nullptr;
    uint32_t *d_wt = nullptr;
    std::vector<unsigned long long> cum(n_types);
    unsigned long long acc = 0;
    for (uint32_t j = 0; j < n_types; ++j) {
        acc += (1ull << 40) / (j + 1ull);
        cum[j] = acc;
    }
    int rc = -1;
    do {
        if (hipMalloc((void **)&d_alpha, alen) != hipSuccess) break;
        if (hipMalloc((void **)&d_tb, (size_t)n_types * 12) != hipSuccess) break;
        if (hipMalloc((void **)&d_tl, n_types) != hipSuccess) break;
        if (hipMalloc((void **)&d_cum, (size_t)n_types * 8) != hipSuccess) break;
        if (hipMalloc((void **)&d_wt, n_cand * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_wl, n_cand) != hipSuccess) break;
        if (hipMalloc((void **)&d_off, (n_cand + 1) * 8) != hipSuccess) break;
        if (hipMalloc((void **)&d_cut, 16) != hipSuccess) break;
        if (hipMemcpy(d_alpha, alphabet, alen, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_cum, cum.data(), (size_t)n_types * 8, hipMemcpyHostToDevice) != hipSuccess) break;
        hipLaunchKernelGGL(k_synth_lexicon, dim3((n_types + 255) / 256), dim3(256), 0, s, seed, n_types, d_alpha, alen, d_tb, d_tl);
        hipLaunchKernelGGL(k_synth_draw, dim3((uint32_t)((n_cand + 255) / 256)), dim3(256), 0, s, seed, n_cand, d_cum, n_types,
                           d_tl, prefix, d_wt, d_wl);
        if (hipGetLastError() != hipSuccess) break;
        if (exclusive_scan<uint8_t>(s, d_wl, n_cand, d_off, n_cand + 1) != 0) break;
        hipLaunchKernelGGL(k_synth_cut, dim3(1), dim3(1), 0, s, d_off, n_cand, target, d_cut);
        unsigned long long cut[2];
        if (hipMemcpyAsync(cut, d_cut, 16, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        if (cut[0] == 0) { rc = -2; break; }
        if (hipMalloc((void **)&d_bytes, cut[1]) != hipSuccess) break;
        hipLaunchKernelGGL(k_synth_fill, dim3((uint32_t)((cut[0] + 255) / 256)), dim3(256), 0, s, d_wt, d_off, cut[0], d_tb, d_tl, prefix, d_bytes);
        if (hipGetLastError() != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        out->bytes = d_bytes;
        out->off = d_off;
        out->n_words = cut[0];
        out->n_bytes = cut[1];
        d_bytes = nullptr;
        d_off = nullptr;
        rc = 0;
    } while (0);
    (void)hipFree(d_alpha); (void)hipFree(d_tb); (void)hipFree(d_tl); (void)hipFree(d_cum);
    (void)hipFree(d_wt); (void)hipFree(d_wl); (void)hipFree(d_cut);
    if (d_bytes) (void)hipFree(d_bytes);
    if (d_off) (void)hipFree(d_off);
    return rc;
}

// ---------------------------------------------------------------- the same Zipf draw over a lexicon the caller supplies
// (yet_another_bpe/synth.py: text_lexicon -- multi-byte UTF-8 words, digits, punctuation, whitespace runs, long runs): the
// "words" drawn are pieces of TEXT, concatenated; the pre-tokeniser decides where the pre-tokens are.
__global__ void k_synth_draw_lex(unsigned long long seed, unsigned long long n_cand, const unsigned long long *cum, uint32_t n_types,
                                 const unsigned long long *lex_off, uint32_t *word_type, uint32_t *word_len) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_cand) return;
    const unsigned long long u = synth_rnd(seed, 3, i) % cum[n_types - 1];
    uint32_t lo = 0, hi = n_types;  // first j with cum[j] > u
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (cum[mid] > u) hi = mid; else lo = mid + 1;
    }
    word_type[i] = lo;
    word_len[i] = (uint32_t)(lex_off[lo + 1] - lex_off[lo]);
}
__global__ void k_synth_fill_lex(const uint32_t *word_type, const unsigned long long *off, unsigned long long n_words,
                                 const uint8_t *lex_bytes, const unsigned long long *lex_off, uint8_t *bytes) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_words) return;
    const uint32_t ty = word_type[i];
    uint8_t *dst = bytes + off[i];
    const uint8_t *src = lex_bytes + lex_off[ty];
    const uint32_t n = (uint32_t)(lex_off[ty + 1] - lex_off[ty]);
    for (uint32_t k = 0; k < n; ++k) dst[k] = src[k];
}

// lex_bytes / lex_off: HOST arrays (n_types + 1 offsets).  n_cand: how many draws to make (>= the words the target needs).
inline int synth_generate_lex(hipStream_t s, unsigned long long target, uint32_t n_types, unsigned long long seed,
                              const uint8_t *lex_bytes, const unsigned long long *lex_off, unsigned long long n_cand, SynthOut *out) {
    uint8_t *d_lb = nullptr, *d_bytes = nullptr;
    unsigned long long *d_lo = nullptr, *d_cum = nullptr, *d_off = nullptr, *d_cut = nullptr;
    uint32_t *d_wt = nullptr, *d_wl = nullptr;
    std::vector<unsigned long long> cum(n_types);
    unsigned long long acc = 0;
    for (uint32_t j = 0; j < n_types; ++j) {
        acc += (1ull << 40) / (j + 1ull);
        cum[j] = acc;
    }
    const unsigned long long lex_total = lex_off[n_types];
    int rc = -1;
    do {
        if (hipMalloc((void **)&d_lb, lex_total ? lex_total : 1) != hipSuccess) break;
        if (hipMalloc((void **)&d_lo, ((size_t)n_types + 1) * 8) != hipSuccess) break;
        if (hipMalloc((void **)&d_cum, (size_t)n_types * 8) != hipSuccess) break;
        if (hipMalloc((void **)&d_wt, n_cand * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_wl, n_cand * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_off, (n_cand + 1) * 8) != hipSuccess) break;
        if (hipMalloc((void **)&d_cut, 16) != hipSuccess) break;
        if (hipMemcpy(d_lb, lex_bytes, lex_total, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_lo, lex_off, ((size_t)n_types + 1) * 8, hipMemcpyHostToDevice) != hipSuccess) break;
        if (hipMemcpy(d_cum, cum.data(), (size_t)n_types * 8, hipMemcpyHostToDevice) != hipSuccess) break;
        hipLaunchKernelGGL(k_synth_draw_lex, dim3((uint32_t)((n_cand + 255) / 256)), dim3(256), 0, s, seed, n_cand, d_cum, n_types, d_lo, d_wt, d_wl);
        if (hipGetLastError() != hipSuccess) break;
        if (exclusive_scan<uint32_t>(s, d_wl, n_cand, d_off, n_cand + 1) != 0) break;
        hipLaunchKernelGGL(k_synth_cut, dim3(1), dim3(1), 0, s, d_off, n_cand, target, d_cut);
        unsigned long long cut[2];
        if (hipMemcpyAsync(cut, d_cut, 16, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        if (cut[0] == 0) { rc = -2; break; }  // (not enough draws for the target)
        if (hipMalloc((void **)&d_bytes, cut[1]) != hipSuccess) break;
        hipLaunchKernelGGL(k_synth_fill_lex, dim3((uint32_t)((cut[0] + 255) / 256)), dim3(256), 0, s, d_wt, d_off, cut[0], d_lb, d_lo, d_bytes);
        if (hipGetLastError() != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        out->bytes = d_bytes;
        out->off = d_off;
        out->n_words = cut[0];
        out->n_bytes = cut[1];
        d_bytes = nullptr;
        d_off = nullptr;
        rc = 0;
    } while (0);
    (void)hipFree(d_lb); (void)hipFree(d_lo); (void)hipFree(d_cum);
    (void)hipFree(d_wt); (void)hipFree(d_wl); (void)hipFree(d_cut);
    if (d_bytes) (void)hipFree(d_bytes);
    if (d_off) (void)hipFree(d_off);
    return rc;
}

// ================================================================ device-side pooling of equal words (trainer.py:221-225)
__global__ void k_word_hash(const uint8_t *bytes, const unsigned long long *off, unsigned long long n, unsigned long long *hash) {
    const unsigned long long w = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const uint8_t *p = bytes + off[w];
    const unsigned long long L = off[w + 1] - off[w];
    unsigned long long h = 0xcbf29ce484222325ull ^ L;
    for (unsigned long long i = 0; i < L; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    hash[w] = synth_mix(h);
}

// rep[w] = index of the representative (first inserter) of w's byte string; count[rep] += freq(w)
// Every word finds (or becomes) the representative of its byte string in an open-addressing set of word indices, and
// adds its frequency to the representative's count.  Word frequencies are Zipfian -- the most common word of a 1 GiB
// corpus occurs ~10 M times -- and same-address HBM atomics serialise (~11 ns each), so the counts are first summed
// per workgroup in a small LDS table: a hot word then costs one HBM atomic per workgroup, not one per occurrence.
constexpr int DEDUP_AGG = 512;
__global__ __launch_bounds__(256) void k_word_dedup(const uint8_t *bytes, const unsigned long long *off, const unsigned long long *freq,
                                                    unsigned long long n, const unsigned long long *hash, uint32_t *slots, unsigned long long mask,
                                                    uint32_t *rep, unsigned long long *count) {
    __shared__ uint32_t s_k[DEDUP_AGG];
    __shared__ unsigned long long s_v[DEDUP_AGG];
    for (int i = threadIdx.x; i < DEDUP_AGG; i += 256) {
        s_k[i] = EMPTY;
        s_v[i] = 0ull;
    }
    __syncthreads();
    const unsigned long long w = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n) {
        const unsigned long long h = hash[w];
        const unsigned long long o0 = off[w], L = off[w + 1] - o0;
        unsigned long long s = h & mask;
        uint32_t r = EMPTY;
        while (true) {
            uint32_t cur = __hip_atomic_load(&slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == EMPTY) {
                cur = atomicCAS(&slots[s], EMPTY, (uint32_t)w);
                if (cur == EMPTY) { r = (uint32_t)w; break; }
            }
            if (hash[cur] == h) {
                const unsigned long long c0 = off[cur];
                if (off[cur + 1] - c0 == L) {
                    bool eq = true;
                    for (unsigned long long i = 0; i < L; ++i)
                        if (bytes[c0 + i] != bytes[o0 + i]) { eq = false; break; }
                    if (eq) { r = cur; break; }
                }
            }
            s = (s + 1) & mask;
        }
        rep[w] = r;
        const unsigned long long f = freq ? freq[w] : 1ull;
        uint32_t a = hash32(r) & (DEDUP_AGG - 1);
        bool done = false;
        for (int probe = 0; probe < 8 && !done; ++probe) {
            uint32_t k = __hip_atomic_load(&s_k[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (k == EMPTY) k = atomicCAS(&s_k[a], EMPTY, r);
            if (k == EMPTY || k == r) {
                atomicAdd(&s_v[a], f);
                done = true;
            }
            a = (a + 1) & (DEDUP_AGG - 1);
        }
        if (!done) atomicAdd(&count[r], f); // the LDS table is crowded around this hash: straight to HBM
    }
    __syncthreads();
    for (int i = threadIdx.x; i < DEDUP_AGG; i += 256)
        if (s_k[i] != EMPTY && s_v[i]) atomicAdd(&count[s_k[i]], s_v[i]);
}

__global__ void k_dedup_flags(const uint32_t *rep, const unsigned long long *off, unsigned long long n, uint32_t *flag, uint32_t *ulen) {
    const unsigned long long w = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    const bool is_rep = rep[w] == (uint32_t)w;
    flag[w] = is_rep ? 1u : 0u;
    ulen[w] = is_rep ? (uint32_t)(off[w + 1] - off[w]) : 0u;
}

__global__ void k_dedup_gather(const uint8_t *bytes, const unsigned long long *off, const uint32_t *rep, unsigned long long n,
                               const unsigned long long *uidx, const unsigned long long *uoff, const unsigned long long *count,
                               uint8_t *out_bytes, unsigned long long *out_off, unsigned long long *out_freq) {
    const unsigned long long w = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n) return;
    if (w == 0) out_off[uidx[n]] = uoff[n];
    if (rep[w] != (uint32_t)w) return;
    const unsigned long long u = uidx[w], o = uoff[w], o0 = off[w], L = off[w + 1] - o0;
    out_off[u] = o;
    out_freq[u] = count[w];
    for (unsigned long long i = 0; i < L; ++i) out_bytes[o + i] = bytes[o0 + i];
}

struct DedupOut {
    uint8_t *bytes;
    unsigned long long *off;
    unsigned long long *freq;
    unsigned long long n_unique, total_bytes;
};

inline int dedup_words(hipStream_t s, const uint8_t *bytes, const unsigned long long *off, const unsigned long long *freq,
                       unsigned long long n, unsigned long long total_bytes, DedupOut *out) {
    (void)total_bytes;
    unsigned long long cap = 1024;
    while (cap < n * 2) cap <<= 1;
    unsigned long long *d_hash = nullptr, *d_count = nullptr, *d_uidx = nullptr, *d_uoff = nullptr;
    uint32_t *d_slots = nullptr, *d_rep = nullptr, *d_flag = nullptr, *d_ulen = nullptr;
    uint8_t *o_bytes = nullptr;
    unsigned long long *o_off = nullptr, *o_freq = nullptr;
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    int rc = -1;
    do {
        if (hipMalloc((void **)&d_hash, n * 8) != hipSuccess) break;
        if (hipMalloc((void **)&d_count, n * 8) != hipSuccess) break;
        if (hipMalloc((void **)&d_slots, cap * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_rep, n * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_flag, n * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_ulen, n * 4) != hipSuccess) break;
        if (hipMalloc((void **)&d_uidx, (n + 1) * 8) != hipSuccess) break;
        if (hipMalloc((void **)&d_uoff, (n + 1) * 8) != hipSuccess) break;
        if (hipMemsetAsync(d_slots, 0xFF, cap * 4, s) != hipSuccess) break;
        if (hipMemsetAsync(d_count, 0, n * 8, s) != hipSuccess) break;
        hipLaunchKernelGGL(k_word_hash, dim3(grid), dim3(256), 0, s, bytes, off, n, d_hash);
        hipLaunchKernelGGL(k_word_dedup, dim3(grid), dim3(256), 0, s, bytes, off, freq, n, d_hash, d_slots, cap - 1, d_rep, d_count);
        hipLaunchKernelGGL(k_dedup_flags, dim3(grid), dim3(256), 0, s, d_rep, off, n, d_flag, d_ulen);
        if (hipGetLastError() != hipSuccess) break;
        if (exclusive_scan<uint32_t>(s, d_flag, n, d_uidx, n + 1) != 0) break;
        if (exclusive_scan<uint32_t>(s, d_ulen, n, d_uoff, n + 1) != 0) break;
        unsigned long long nu = 0, tb = 0;
        if (hipMemcpyAsync(&nu, d_uidx + n, 8, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipMemcpyAsync(&tb, d_uoff + n, 8, hipMemcpyDeviceToHost, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        if (hipMalloc((void **)&o_bytes, tb ? tb : 1) != hipSuccess) break;
        if (hipMalloc((void **)&o_off, (nu + 1) * 8) != hipSuccess) break;
        if (hipMalloc((void **)&o_freq, (nu ? nu : 1) * 8) != hipSuccess) break;
        hipLaunchKernelGGL(k_dedup_gather, dim3(grid), dim3(256), 0, s, bytes, off, d_rep, n, d_uidx, d_uoff, d_count, o_bytes, o_off, o_freq);
        if (hipGetLastError() != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        out->bytes = o_bytes; out->off = o_off; out->freq = o_freq;
        out->n_unique = nu; out->total_bytes = tb;
        o_bytes = nullptr; o_off = nullptr; o_freq = nullptr;
        rc = 0;
    } while (0);
    (void)hipFree(d_hash); (void)hipFree(d_count); (void)hipFree(d_slots); (void)hipFree(d_rep);
    (void)hipFree(d_flag); (void)hipFree(d_ulen); (void)hipFree(d_uidx); (void)hipFree(d_uoff);
    if (o_bytes) (void)hipFree(o_bytes);
    if (o_off) (void)hipFree(o_off);
    if (o_freq) (void)hipFree(o_freq);
    return rc;
}

// ================================================================ retile (flat layout): repack live words into fresh tiles
struct NoMerge {
    __device__ __forceinline__ int operator()(int) const { return 0; }
};

struct RetileParams {
    const uint16_t *tiles;
    const uint32_t *tile_len;
    uint32_t n_tiles;
    uint32_t *kept;                  // pass 1 out
    const unsigned long long *base;  // pass 2 in: exclusive scan of kept
    uint16_t *new_tiles;
    uint32_t *new_len;
};

template <bool SCATTER>
__global__ __launch_bounds__(BLOCK) void k_retile(RetileParams P) {
    __shared__ __attribute__((aligned(16))) uint16_t s_stage[WPB][8 + CAP + 8];
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint16_t *stg = s_stage[wib];
    if (lane < 8) {
        stg[lane] = YB_PAD;
        stg[8 + CAP + lane] = YB_PAD;
    }
    __syncthreads();
    TokAt T{stg};
    NoMerge M;
    const uint32_t stride = gridDim.x * WPB;
    for (uint32_t tile = blockIdx.x * WPB + wib; tile < P.n_tiles; tile += stride) {
        const uint32_t len = P.tile_len[tile];
        if (len == 0) {
            if (!SCATTER && lane == 0) P.kept[tile] = 0;
            continue;
        }
        TileRegs r = load_tile(P.tiles, tile, len, lane);
        wave_sync();
        stage_tile(stg, r, lane);
        wave_sync();
        const int rounds = (len + 63) >> 6;
        uint32_t outpos = 0;
        uint32_t carry_ws = 0;  // kept-index where the current word starts
        const unsigned long long base = SCATTER ? P.base[tile] : 0ull;
        for (int k = 0; k < rounds; ++k) {
            const int p = k * 64 + lane;
            uint32_t o = 0;
            const int keep = yb_keep(p, 0u, true, T, M, o);
            const unsigned long long km = __ballot(keep);
            if (SCATTER) {
                const unsigned long long sm = __ballot(keep && o == YB_SEP);
                const uint32_t kidx = outpos + __popcll(km & lanemask_lt(lane));
                const unsigned long long lower = sm & lanemask_lt(lane);
                uint32_t ws = carry_ws;
                if (lower) {
                    const int j = 63 - __clzll((long long)lower);
                    ws = outpos + __popcll(km & ((j == 63) ? ~0ull : ((2ull << j) - 1ull)));
                }
                if (keep) {
                    const unsigned long long g = base + kidx, gws = base + ws;
                    const unsigned long long nt = gws / SPAN;
                    const uint32_t slot = (uint32_t)(g - nt * SPAN);
                    P.new_tiles[nt * CAP + slot] = (uint16_t)o;
                    if (o == YB_SEP) atomicMax(&P.new_len[nt], slot + 1);  // one per word
                }
                if (sm) {
                    const int j = 63 - __clzll((long long)sm);
                    carry_ws = outpos + __popcll(km & ((j == 63) ? ~0ull : ((2ull << j) - 1ull)));
                }
            }
            outpos += __popcll(km);
        }
        if (!SCATTER && lane == 0) P.kept[tile] = outpos;
    }
}

// ================================================================ latency probe (measurement; include/yabpe.h yabpe_latency_probe)
// The pieces the per-merge launch is built from, measured on an otherwise idle device: one lane walks a chain of dependent
// accesses through a table far larger than the caches, so every hop is a full trip to memory.  The chain is the full-period
// affine map x -> (a x + c) mod 2^k (a = 1 mod 4, c odd: one cycle over all entries): consecutive hops land on unrelated
// lines AND unrelated pages, like the scattered accesses of the launch (a constant stride -- what this probe used until
// round 3 -- makes every hop a TLB miss of the same kind and overstates the trip by that much).
//   mode 0: plain loads   mode 1: device-scope loads (past the caches; what hand-offs inside a launch use)
//   mode 2: returning device-scope atomic adds (the adds of the aggregator flush whose results the selection needs)
__global__ void k_chain_init(uint32_t *next, uint32_t n, uint32_t salt) { // n a power of two; next[i] = (a i + c) mod n
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t a = 1664525u, c = 1013904223u + 2u * salt; // (a = 1 mod 4, c odd)
    if (i < n) next[i] = (a * i + c) & (n - 1u);
}
__global__ void k_chain_walk(uint32_t *next, uint32_t hops, int mode, unsigned long long *out_ticks, uint32_t *sink) {
    uint32_t x = 0;
    const unsigned long long t0 = wall_clock64();
    for (uint32_t h = 0; h < hops; ++h) {
        if (mode == 0) x = next[x];
        else if (mode == 1) x = __hip_atomic_load(&next[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else x = atomicAdd(&next[x], 0u);
    }
    const unsigned long long t1 = wall_clock64();
    *out_ticks = t1 - t0; // 100 MHz
    *sink = x;
}
__global__ void k_empty(uint32_t *sink) {
    if (threadIdx.x == 12345u) *sink = 1;
}

}  // namespace yb
