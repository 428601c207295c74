// Dev tool: what does a COLD instruction cache cost a short kernel on gfx950?  The per-merge kernels run ~20 KB of
// straight-line code once per wave per launch.  k_straight<N> executes N KB of distinct VALU code once; k_loop executes the
// same number of instructions as a small loop (code resident after the first iteration).  Each workgroup stamps the 100 MHz
// wall clock around its body; launches are back to back on one stream, so every launch starts cold the way the real ones do.
//   hipcc --offload-arch=gfx950 -O3 -o tools/icache_microbench tools/icache_microbench.hip && ./tools/icache_microbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

// 64 dependent-free VALU instructions (8 B each with a literal) = 512 B of code per block
#define I8(a) "v_xor_b32 %0, 0x" #a "1, %0\n v_add_u32 %0, 0x" #a "3, %0\n v_xor_b32 %0, 0x" #a "5, %0\n v_add_u32 %0, 0x" #a "7, %0\n" \
              "v_xor_b32 %0, 0x" #a "9, %0\n v_add_u32 %0, 0x" #a "b, %0\n v_xor_b32 %0, 0x" #a "d, %0\n v_add_u32 %0, 0x" #a "f, %0\n"
#define B512(x) asm volatile(I8(1234) I8(2345) I8(3456) I8(4567) I8(5678) I8(6789) I8(789a) I8(89ab) : "+v"(x));
#define KB1(x) B512(x) B512(x)
#define KB4(x) KB1(x) KB1(x) KB1(x) KB1(x)
#define KB16(x) KB4(x) KB4(x) KB4(x) KB4(x)

template <int KB>
__global__ void k_straight(unsigned long long *stamps, uint32_t *out) {
    const unsigned long long t0 = wall_clock64();
    uint32_t x = threadIdx.x;
    if constexpr (KB >= 16) { KB16(x) }
    if constexpr (KB >= 32) { KB16(x) }
    if constexpr (KB >= 48) { KB16(x) }
    if constexpr (KB == 4) { KB4(x) }
    if (x == 0xdeadbeef) out[0] = x;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}
__global__ void k_loop(unsigned long long *stamps, uint32_t *out, int iters) { // iters x 512 B worth of instructions
    const unsigned long long t0 = wall_clock64();
    uint32_t x = threadIdx.x;
    for (int i = 0; i < iters; ++i) { B512(x) }
    if (x == 0xdeadbeef) out[0] = x;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t0; stamps[2 * blockIdx.x + 1] = wall_clock64(); }
}

template <class F>
static void run(const char *name, int grid, unsigned long long *d_st, F launch) {
    std::vector<unsigned long long> h(2 * grid);
    for (int rep = 0; rep < 3; ++rep) launch();
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    for (int rep = 0; rep < 50; ++rep) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> du(grid);
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int i = 0; i < grid; ++i) { t0 = std::min(t0, h[2 * i]); t1 = std::max(t1, h[2 * i + 1]); du[i] = (h[2 * i + 1] - h[2 * i]) / 100.0; }
    std::sort(du.begin(), du.end());
    printf("%-28s grid %4d: body p50 %6.2f max %6.2f us | first start -> last end %6.2f us | back-to-back %6.2f us/launch\n", name, grid, du[grid / 2], du[grid - 1], (t1 - t0) / 100.0, ms * 1000 / 50);
}

int main() {
    unsigned long long *d_st; uint32_t *d_out;
    hipMalloc(&d_st, 2 * 4096 * 8); hipMalloc(&d_out, 64);
    for (int grid : {256, 768}) {
        run("empty (4 KB straight)", grid, d_st, [&] { hipLaunchKernelGGL(k_straight<4>, dim3(grid), dim3(256), 0, 0, d_st, d_out); });
        run("16 KB straight-line", grid, d_st, [&] { hipLaunchKernelGGL(k_straight<16>, dim3(grid), dim3(256), 0, 0, d_st, d_out); });
        run("32 KB straight-line", grid, d_st, [&] { hipLaunchKernelGGL(k_straight<32>, dim3(grid), dim3(256), 0, 0, d_st, d_out); });
        run("48 KB straight-line", grid, d_st, [&] { hipLaunchKernelGGL(k_straight<48>, dim3(grid), dim3(256), 0, 0, d_st, d_out); });
        run("loop = 16 KB of instrs", grid, d_st, [&] { hipLaunchKernelGGL(k_loop, dim3(grid), dim3(256), 0, 0, d_st, d_out, 32); });
        run("loop = 32 KB of instrs", grid, d_st, [&] { hipLaunchKernelGGL(k_loop, dim3(grid), dim3(256), 0, 0, d_st, d_out, 64); });
        run("loop = 48 KB of instrs", grid, d_st, [&] { hipLaunchKernelGGL(k_loop, dim3(grid), dim3(256), 0, 0, d_st, d_out, 96); });
    }
    return 0;
}
