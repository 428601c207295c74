// yabpe_pretok_kernels.h -- the pre-tokeniser on the device (reference trainer.py:136-214).
//
// Input: the UTF-8 bytes of the corpus, resident in HBM, cut into chunks (every chunk is a text of its own).
// Output: u64 word offsets INTO THAT BUFFER (no copy of the text): word i = text[off[i], off[i+1]).  Pre-tokens partition
// each chunk, so one flag per byte ("a pre-token starts here") describes the result; the flags are compacted into the
// offsets array by a two-level prefix sum.  All rules live in pretok_logic.h (shared with the CPU model the tests run
// against regex.findall); the kernels only map bytes to threads.
//
// Passes over the text:
//               k_pt_fused     text -> meta (class, continuation, UTF-8 validation: first malformed byte by atomicMin)
//                                      and flags, through an LDS window per workgroup
//               k_pt_special   text, meta -> corrected flags (only when special tokens are configured)
//               k_pt_count / k_pt_scatter   flags -> offsets
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pretok_logic.h"
#include "yabpe_aux_kernels.h"

namespace yb {

constexpr int PT_PER_BLOCK = BLOCK * 8; // bytes per workgroup in the count / scatter passes

struct PretokParams {
    const uint8_t *text;
    uint8_t *meta;
    uint8_t *flags;
    unsigned long long n;
    const uint8_t *cls;       // class per code point (0x110000 entries)
    unsigned long long *err;  // smallest malformed byte position (atomicMin), ~0 = none
    PtSpecials sp;
};

__global__ __launch_bounds__(BLOCK) void k_pt_mark_chunks(uint8_t *meta, const unsigned long long *chunk_off, uint32_t n_chunks,
                                                          unsigned long long n) {
    const uint32_t c = blockIdx.x * BLOCK + threadIdx.x;
    if (c < n_chunks && chunk_off[c] < n) meta[chunk_off[c]] = PT_CHUNK0; // (meta was zeroed)
}

// classify + starts in one pass: a workgroup stages a window of the text (and the chunk marks) in LDS with a halo on both
// sides, classifies every byte of window + halo there (neighbouring workgroups redo each other's halo: 16 bytes of 4 KiB),
// then decides the start flag of every byte of the window from LDS.  The text is read from HBM once, meta and flags are
// written once, 16 B per lane.  The rules are the same functions the CPU model runs, on an LDS view (PtView::org).
constexpr int PT_WIN = BLOCK * 16; // bytes per workgroup and iteration
constexpr int PT_HALO = 16;        // >= reach of the rules: 7 bytes back (contraction + previous character), 4 ahead
constexpr int PT_LDS = PT_HALO + PT_WIN + PT_HALO;
__global__ __launch_bounds__(BLOCK) void k_pt_fused(PretokParams P) {
    __shared__ __attribute__((aligned(16))) uint8_t s_text[PT_LDS];
    __shared__ __attribute__((aligned(16))) uint8_t s_meta[PT_LDS];
    __shared__ __attribute__((aligned(16))) uint8_t s_tmp[PT_LDS]; // classes, then flags
    const unsigned long long n_win = (P.n + PT_WIN - 1) / PT_WIN;
    const bool wide = (reinterpret_cast<uintptr_t>(P.text) & 15u) == 0; // 16-B loads need an aligned text buffer
    for (unsigned long long w = blockIdx.x; w < n_win; w += gridDim.x) {
        const unsigned long long base = w * PT_WIN;      // first byte of the window
        const long long org = (long long)base - PT_HALO; // position of s_text[0] (negative for window 0)
        // ---- stage: the window as 16-B pieces, the halos byte by byte
        {
            const unsigned long long g = base + (unsigned long long)threadIdx.x * 16;
            uint8_t *dt = s_text + PT_HALO + threadIdx.x * 16, *dm = s_meta + PT_HALO + threadIdx.x * 16;
            if (g + 16 <= P.n) {
                if (wide) {
                    *reinterpret_cast<uint4 *>(dt) = *reinterpret_cast<const uint4 *>(P.text + g);
                } else {
                    for (int k = 0; k < 16; ++k) dt[k] = P.text[g + k];
                }
                *reinterpret_cast<uint4 *>(dm) = *reinterpret_cast<const uint4 *>(P.meta + g);
            } else {
                for (int k = 0; k < 16; ++k) {
                    dt[k] = g + k < P.n ? P.text[g + k] : 0;
                    dm[k] = g + k < P.n ? P.meta[g + k] : 0;
                }
            }
            if (threadIdx.x < 2 * PT_HALO) {
                const int k = threadIdx.x < PT_HALO ? threadIdx.x : PT_WIN + threadIdx.x; // slot in s_text
                const long long pos = org + k;
                const bool in = pos >= 0 && (unsigned long long)pos < P.n;
                s_text[k] = in ? P.text[pos] : 0;
                s_meta[k] = in ? P.meta[pos] : 0;
            }
        }
        __syncthreads();
        // A view whose position `org` is s_text[0].  For window 0 the halo before the text does not exist: the view then
        // starts at position 0 (s_text + PT_HALO) -- no rule looks before a chunk start, and position 0 is one.
        const bool first = org < 0;
        const unsigned long long vorg = first ? 0ull : (unsigned long long)org;
        const PtView v{first ? s_text + PT_HALO : s_text, first ? s_meta + PT_HALO : s_meta, P.n, vorg};
        uint8_t *tmp = first ? s_tmp + PT_HALO : s_tmp;   // tmp[pos - vorg]
        uint8_t *meta_w = first ? s_meta + PT_HALO : s_meta;
        // ---- classify the window + 8 bytes on each side (what the start rules can look at)
        const long long c_lo = (long long)base - 8, c_hi = (long long)base + PT_WIN + 8; // [c_lo, c_hi)
        {
            unsigned long long bad_pos = ~0ull;
            for (long long pos = c_lo + threadIdx.x; pos < c_hi; pos += BLOCK) { // (consecutive lanes, consecutive bytes)
                if (pos < 0 || (unsigned long long)pos >= P.n) continue;
                const unsigned long long i = (unsigned long long)pos;
                unsigned long long end = P.n;
                for (unsigned long long q = i + 1; q < i + 4 && q < P.n; ++q)
                    if (v.M(q) & PT_CHUNK0) {
                        end = q;
                        break;
                    }
                bool bad = false;
                tmp[i - vorg] = pt_classify(v, i, end, P.cls, &bad);
                if (bad && i >= base && i < base + PT_WIN && i < bad_pos) bad_pos = i; // (a halo byte is reported by its own window)
            }
            if (bad_pos != ~0ull) atomicMin(P.err, bad_pos);
        }
        __syncthreads(); // every chunk mark has been read; now the class bits are added
        for (long long pos = c_lo + threadIdx.x; pos < c_hi; pos += BLOCK) {
            if (pos < 0 || (unsigned long long)pos >= P.n) continue;
            const unsigned long long k = (unsigned long long)pos - vorg;
            meta_w[k] = (uint8_t)((meta_w[k] & PT_CHUNK0) | tmp[k]);
        }
        __syncthreads();
        // ---- start flags of the window (tmp is free again), then meta and flags go out as 16-B pieces
        for (int k = threadIdx.x; k < PT_WIN; k += BLOCK) {
            const unsigned long long j = base + k;
            tmp[j - vorg] = j < P.n ? (pt_is_start(v, j, -1) ? 1 : 0) : 0;
        }
        __syncthreads();
        {
            const unsigned long long g = base + (unsigned long long)threadIdx.x * 16;
            const uint8_t *fm = s_meta + PT_HALO + threadIdx.x * 16, *ff = s_tmp + PT_HALO + threadIdx.x * 16;
            if (g + 16 <= P.n) {
                *reinterpret_cast<uint4 *>(P.flags + g) = *reinterpret_cast<const uint4 *>(ff);
                *reinterpret_cast<uint4 *>(P.meta + g) = *reinterpret_cast<const uint4 *>(fm);
            } else {
                for (int k = 0; k < 16 && g + k < P.n; ++k) {
                    P.flags[g + k] = ff[k];
                    P.meta[g + k] = fm[k];
                }
            }
        }
        __syncthreads();
    }
}

// Special tokens: one pass.  A 256-bit set of the specials' first bytes keeps nearly every thread out of the compare;
// a thread that finds an occurrence checks whether it heads its chain (occurrences before it are looked up on demand,
// through the same filter) and, if so, resolves the whole chain (pretok_logic.h).
__global__ __launch_bounds__(BLOCK) void k_pt_special(PretokParams P) {
    __shared__ uint32_t s_first[8];
    if (threadIdx.x < 8) s_first[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x < P.sp.n) {
        const uint32_t o = P.sp.off[threadIdx.x];
        if (P.sp.off[threadIdx.x + 1] > o) atomicOr(&s_first[P.sp.bytes[o] >> 5], 1u << (P.sp.bytes[o] & 31));
    }
    __syncthreads();
    const PtView v{P.text, P.meta, P.n, 0};
    const PtSpecials sp = P.sp;
    const uint32_t *first = s_first;
    auto occ = [&](unsigned long long q) -> uint32_t {
        const uint8_t b = v.T(q);
        return ((first[b >> 5] >> (b & 31)) & 1u) ? pt_special_at(v, sp, q) : 0u;
    };
    for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < P.n; i += (unsigned long long)gridDim.x * BLOCK) {
        const uint32_t o = occ(i);
        if (o && pt_special_is_head(v, sp, occ, i)) pt_special_walk(v, sp, occ, P.flags, i, o);
    }
}

// flags -> offsets, pass 1: number of starts per workgroup of PT_PER_BLOCK bytes
__global__ __launch_bounds__(BLOCK) void k_pt_count(const uint8_t *flags, unsigned long long n, unsigned long long *block_sums) {
    __shared__ uint32_t s_w[WPB];
    const unsigned long long base = (unsigned long long)blockIdx.x * PT_PER_BLOCK + (unsigned long long)threadIdx.x * 8;
    uint32_t cnt = 0;
    if (base + 8 <= n) {
        const unsigned long long w = *reinterpret_cast<const unsigned long long *>(flags + base); // eight 0/1 bytes
        cnt = (uint32_t)__popcll(w);
    } else {
        for (unsigned long long k = base; k < n; ++k) cnt += flags[k];
    }
    cnt = (uint32_t)wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < WPB; ++w) t += s_w[w];
        block_sums[blockIdx.x] = t;
    }
}

// pass 2: off[block_base + rank of the start inside the workgroup] = its byte position
__global__ __launch_bounds__(BLOCK) void k_pt_scatter(const uint8_t *flags, unsigned long long n, const unsigned long long *block_base,
                                                      unsigned long long *off) {
    __shared__ uint32_t s_w[WPB];
    const unsigned long long base = (unsigned long long)blockIdx.x * PT_PER_BLOCK + (unsigned long long)threadIdx.x * 8;
    uint8_t f[8];
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f[k] = base + k < n ? flags[base + k] : 0;
        cnt += f[k];
    }
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_sum(cnt);
    if (lane == 63) s_w[wib] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wib; ++w) woff += s_w[w];
    unsigned long long idx = block_base[blockIdx.x] + woff + inc - cnt;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (f[k]) off[idx++] = base + k;
}

struct PretokOut {
    unsigned long long *off; // n_words + 1
    unsigned long long n_words;
    long long bad_pos;       // first malformed UTF-8 byte, -1 if the text is valid
};

// Runs all passes on `text` (device).  chunk_off: device array of n_chunks chunk starts.  cls: device class table.
// Scratch (meta, flags) is allocated and released here; out->off is the caller's to free.
inline int pretokenize(hipStream_t s, const uint8_t *text, unsigned long long n, const unsigned long long *chunk_off, uint32_t n_chunks,
                       const uint8_t *cls, const PtSpecials &sp_dev, PretokOut *out) {
    out->off = nullptr;
    out->n_words = 0;
    out->bad_pos = -1;
    if (n == 0) {
        YB_RET(hipMalloc((void **)&out->off, 8));
        YB_RET(hipMemsetAsync(out->off, 0, 8, s));
        YB_RET(hipStreamSynchronize(s));
        return 0;
    }
    uint8_t *meta = nullptr, *flags = nullptr;
    unsigned long long *err = nullptr, *sums = nullptr, *bases = nullptr;
    const unsigned long long nb = (n + PT_PER_BLOCK - 1) / PT_PER_BLOCK;
    int rc = -1;
    do {
        if (hipMalloc((void **)&meta, n) != hipSuccess || hipMalloc((void **)&flags, n + 8) != hipSuccess) break;
        if (hipMalloc((void **)&err, 8) != hipSuccess || hipMalloc((void **)&sums, nb * 8) != hipSuccess) break;
        if (hipMalloc((void **)&bases, (nb + 1) * 8) != hipSuccess) break;
        if (hipMemsetAsync(meta, 0, n, s) != hipSuccess || hipMemsetAsync(err, 0xff, 8, s) != hipSuccess) break;
        const uint32_t grid = (uint32_t)std::min<unsigned long long>((n + BLOCK - 1) / BLOCK, 1u << 20);
        PretokParams P{text, meta, flags, n, cls, err, sp_dev};
        hipLaunchKernelGGL(k_pt_mark_chunks, dim3((n_chunks + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, meta, chunk_off, n_chunks, n);
        const uint32_t wgrid = (uint32_t)std::min<unsigned long long>((n + PT_WIN - 1) / PT_WIN, 1u << 20);
        hipLaunchKernelGGL(k_pt_fused, dim3(wgrid), dim3(BLOCK), 0, s, P);
        unsigned long long h_err = 0;
        if (hipMemcpyAsync(&h_err, err, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) break;
        if (h_err != ~0ull) { // malformed UTF-8: nothing else is computed (the neighbour walks assume valid text)
            out->bad_pos = (long long)h_err;
            rc = 0;
            break;
        }
        if (sp_dev.n) hipLaunchKernelGGL(k_pt_special, dim3(grid), dim3(BLOCK), 0, s, P);
        hipLaunchKernelGGL(k_pt_count, dim3((uint32_t)nb), dim3(BLOCK), 0, s, flags, n, sums);
        if (hipGetLastError() != hipSuccess) break;
        if (exclusive_scan<unsigned long long>(s, sums, nb, bases, nb + 1) != 0) break;
        unsigned long long total = 0;
        if (hipMemcpyAsync(&total, bases + nb, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) break;
        if (hipMalloc((void **)&out->off, (total + 1) * 8) != hipSuccess) break;
        hipLaunchKernelGGL(k_pt_scatter, dim3((uint32_t)nb), dim3(BLOCK), 0, s, flags, n, bases, out->off);
        if (hipMemcpyAsync(out->off + total, &n, 8, hipMemcpyHostToDevice, s) != hipSuccess) break;
        if (hipStreamSynchronize(s) != hipSuccess) break;
        out->n_words = total;
        rc = 0;
    } while (false);
    (void)hipFree(meta);
    (void)hipFree(flags);
    (void)hipFree(err);
    (void)hipFree(sums);
    (void)hipFree(bases);
    if (rc != 0 && out->off) {
        (void)hipFree(out->off);
        out->off = nullptr;
    }
    return rc;
}

} // namespace yb
