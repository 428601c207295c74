// Dev tool: how fast does gfx950 serve device-scope atomics that many workgroups aim at the SAME few addresses (the hot
// pair counts of a sparse merge)?  `grid` workgroups; in each, `per_wg` lanes add to `hot` distinct 64-bit counters (lane i
// -> counter i % hot, on separate 128-B lines) -- returning (the value is used) or not -- and the workgroup stamps how long
// it waited.  Prints the mean / max wait per workgroup and the span from the first start to the last end.
//   hipcc --offload-arch=gfx950 -O3 -o tools/atomic_microbench tools/atomic_microbench.hip && ./tools/atomic_microbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <bool RET, bool U32>
__global__ void k_hot(unsigned long long *ctr, unsigned long long *stamps, int hot, int per_wg, unsigned long long *sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long got = 0;
    if ((int)threadIdx.x < per_wg) {
        unsigned long long *p = ctr + (size_t)(threadIdx.x % hot) * 16;
        if (U32) {
            if (RET) got = atomicAdd(reinterpret_cast<unsigned int *>(p), 1u);
            else atomicAdd(reinterpret_cast<unsigned int *>(p), 1u);
        } else {
            if (RET) got = atomicAdd(p, 1ull);
            else atomicAdd(p, 1ull);
        }
    }
    if (RET && got == 0xdeadbeefdeadbeefull) sink[0] = got;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = wall_clock64();
    }
}

template <bool RET, bool U32>
static void run(int grid, int hot, int per_wg, unsigned long long *ctr, unsigned long long *d_st, unsigned long long *sink) {
    std::vector<unsigned long long> h(2 * grid);
    double mean = 0, mx = 0, span = 0;
    const int reps = 10;
    for (int rep = 0; rep < reps + 2; ++rep) {
        hipLaunchKernelGGL((k_hot<RET, U32>), dim3(grid), dim3(64), 0, 0, ctr, d_st, hot, per_wg, sink);
        hipDeviceSynchronize();
        if (rep < 2) continue;
        hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        double s = 0, m = 0;
        for (int i = 0; i < grid; ++i) {
            t0 = std::min(t0, h[2 * i]);
            t1 = std::max(t1, h[2 * i + 1]);
            const double d = (h[2 * i + 1] - h[2 * i]) / 100.0;
            s += d;
            m = std::max(m, d);
        }
        mean += s / grid;
        mx += m;
        span += (t1 - t0) / 100.0;
    }
    printf("%s %s grid %4d hot %3d lanes/wg %2d: wait mean %6.2f max %6.2f us | first start -> last end %6.2f us | %.1f ns per atomic per address\n", RET ? "returning" : "no-return",
           U32 ? "u32" : "u64", grid, hot, per_wg, mean / reps, mx / reps, span / reps, 1000.0 * (span / reps) / ((double)grid * per_wg / hot));
}

int main() {
    unsigned long long *ctr, *d_st, *sink;
    hipMalloc(&ctr, 1 << 20);
    hipMemset(ctr, 0, 1 << 20);
    hipMalloc(&d_st, 2 * 4096 * 8);
    hipMalloc(&sink, 8);
    for (int grid : {448, 224, 56}) {
        for (int hot : {1, 4}) {
            run<true, false>(grid, hot, hot, ctr, d_st, sink);
            run<false, false>(grid, hot, hot, ctr, d_st, sink);
            run<true, true>(grid, hot, hot, ctr, d_st, sink);
            run<false, true>(grid, hot, hot, ctr, d_st, sink);
        }
    }
    run<true, false>(448, 16, 16, ctr, d_st, sink);
    run<true, false>(448, 64, 16, ctr, d_st, sink);
    return 0;
}
