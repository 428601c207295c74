// Dev tool: how fast does gfx950 start the workgroups of one launch?  Every workgroup stamps the 100 MHz wall clock when
// its first thread runs and again after a dependent chain of `hops` global loads (a stand-in for a short latency-bound
// body).  Prints when the first / median / last workgroup started and ended, relative to the first start.
//   hipcc --offload-arch=gfx950 -O3 -o tools/dispatch_microbench tools/dispatch_microbench.hip && ./tools/dispatch_microbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int LDS_BYTES>
__global__ void k_probe(unsigned long long *stamps, const uint32_t *chain, int hops) {
    __shared__ uint32_t lds[LDS_BYTES / 4];
    const unsigned long long t0 = wall_clock64();
    uint32_t x = threadIdx.x + blockIdx.x * blockDim.x;
    lds[threadIdx.x % (LDS_BYTES / 4)] = x;
    for (int h = 0; h < hops; ++h) x = chain[(x * 2654435761u) >> 8]; // 16M-entry table: every hop misses L2 mostly
    __syncthreads();
    if (x == 0xdeadbeef) lds[0] = 1;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = wall_clock64() + (lds[0] == 0xffffffffu);
    }
}

template <int LDS_BYTES>
static void run(int grid, int block, int hops, unsigned long long *d_st, const uint32_t *d_chain) {
    std::vector<unsigned long long> h(2 * grid);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_probe<LDS_BYTES>, dim3(grid), dim3(block), 0, 0, d_st, d_chain, hops);
        hipDeviceSynchronize();
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k_probe<LDS_BYTES>, dim3(grid), dim3(block), 0, 0, d_st, d_chain, hops);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> st(grid), en(grid), du(grid);
    unsigned long long t0 = ~0ull;
    for (int i = 0; i < grid; ++i) t0 = std::min(t0, h[2 * i]);
    for (int i = 0; i < grid; ++i) {
        st[i] = (h[2 * i] - t0) / 100.0;
        en[i] = (h[2 * i + 1] - t0) / 100.0;
        du[i] = en[i] - st[i];
    }
    std::sort(st.begin(), st.end());
    std::sort(en.begin(), en.end());
    std::sort(du.begin(), du.end());
    printf("grid %5d block %4d lds %6d hops %d: start p50 %6.2f p90 %6.2f last %6.2f | end p50 %6.2f last %6.2f | body p50 %5.2f max %5.2f | back-to-back %.2f us/launch\n",
           grid, block, LDS_BYTES, hops, st[grid / 2], st[grid * 9 / 10], st[grid - 1], en[grid / 2], en[grid - 1], du[grid / 2], du[grid - 1], ms * 1000 / 20);
}

int main() {
    unsigned long long *d_st;
    uint32_t *d_chain;
    const size_t N = 1u << 24;
    hipMalloc(&d_st, 2 * 16384 * 8);
    hipMalloc(&d_chain, N * 4);
    std::vector<uint32_t> c(N);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < N; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        c[i] = (uint32_t)s;
    }
    hipMemcpy(d_chain, c.data(), N * 4, hipMemcpyHostToDevice);
    for (int hops : {0, 4}) {
        run<1024>(256, 256, hops, d_st, d_chain);
        run<1024>(448, 256, hops, d_st, d_chain);
        run<1024>(1792, 256, hops, d_st, d_chain);
        run<12288>(1792, 256, hops, d_st, d_chain);
        run<12288>(3584, 256, hops, d_st, d_chain);
        run<12288>(448, 1024, hops, d_st, d_chain);
        run<12288>(896, 512, hops, d_st, d_chain);
        run<1024>(7168, 64, hops, d_st, d_chain);
    }
    return 0;
}
