// Dev tool: does global_load_lds_dwordx4 (gfx950 LDS-DMA) put lane i's 16 bytes at LDS base + 16 i, also under a partial exec
// mask (inactive lanes leave their piece untouched)?   hipcc --offload-arch=gfx950 -O3 -o tools/ldsdma_test tools/ldsdma_test.hip && ./tools/ldsdma_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k(const uint4 *src, uint4 *dst, uint32_t len) {
    __shared__ __attribute__((aligned(16))) uint4 buf[128];
    const int lane = threadIdx.x & 63;
    buf[lane] = make_uint4(0xdeadbeefu, 0, 0, 0);
    buf[64 + lane] = make_uint4(0xdeadbeefu, 0, 0, 0);
    __syncthreads();
    if ((uint32_t)lane * 8u < len) __builtin_amdgcn_global_load_lds(src + lane, &buf[0], 16, 0, 0);
    if (512u + (uint32_t)lane * 8u < len) __builtin_amdgcn_global_load_lds(src + 64 + lane, &buf[64], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __syncthreads();
    dst[lane] = buf[lane];
    dst[64 + lane] = buf[64 + lane];
}
int main() {
    std::vector<uint4> h(128), o(128);
    for (int i = 0; i < 128; ++i) h[i] = make_uint4(i * 4, i * 4 + 1, i * 4 + 2, i * 4 + 3);
    uint4 *d, *e;
    hipMalloc(&d, 128 * 16); hipMalloc(&e, 128 * 16);
    hipMemcpy(d, h.data(), 128 * 16, hipMemcpyHostToDevice);
    int bad = 0;
    for (uint32_t len : {1024u, 1000u, 520u, 512u, 300u, 8u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, len);
        hipMemcpy(o.data(), e, 128 * 16, hipMemcpyDeviceToHost);
        for (int i = 0; i < 128; ++i) {
            const bool live = (uint32_t)(i * 8) < len;
            const bool ok = live ? (o[i].x == (uint32_t)i * 4 && o[i].w == (uint32_t)i * 4 + 3) : o[i].x == 0xdeadbeefu;
            if (!ok) { if (bad < 8) printf("len %u piece %d: got %u %u %u %u\n", len, i, o[i].x, o[i].y, o[i].z, o[i].w); ++bad; }
        }
    }
    printf(bad ? "FAILED (%d)\n" : "ok: lane i -> LDS base + 16 i, masked lanes untouched\n", bad);
    return bad != 0;
}
