// scan_microbench.hip -- dev tool: how fast can the k_apply FAST PATH structure stream tiles on gfx950?
// Variants: prefetch depth (tiles in flight per wave), static LDS per block (occupancy), grid size.
//   hipcc --offload-arch=gfx950 -O3 -o scan_microbench scan_microbench.hip && ./scan_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int CAP = 1024;
constexpr uint32_t PADPAD = 0xFFFEFFFEu;

struct TR { uint4 a, b; };
__device__ __forceinline__ TR ld(const uint16_t* tiles, uint32_t tile, uint32_t len, int lane) {
    const uint4* base = reinterpret_cast<const uint4*>(tiles + (size_t)tile * CAP);
    TR r; r.a = make_uint4(PADPAD, PADPAD, PADPAD, PADPAD); r.b = r.a;
    if ((uint32_t)(lane * 8) < len) r.a = base[lane];
    if ((uint32_t)(512 + lane * 8) < len) r.b = base[64 + lane];
    return r;
}
__device__ __forceinline__ uint32_t next_lane(uint32_t v, uint32_t fill) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ bool match4(uint4 v, uint32_t nxt, uint32_t mk) {
    int m = (v.x == mk) | (v.y == mk) | (v.z == mk) | (v.w == mk);
    m |= (int)(__builtin_amdgcn_alignbit(v.y, v.x, 16) == mk);
    m |= (int)(__builtin_amdgcn_alignbit(v.z, v.y, 16) == mk);
    m |= (int)(__builtin_amdgcn_alignbit(v.w, v.z, 16) == mk);
    m |= (int)(__builtin_amdgcn_alignbit(nxt, v.w, 16) == mk);
    return m != 0;
}
__device__ __forceinline__ bool check(const TR& r, uint32_t mk) {
    const uint32_t b0 = __builtin_amdgcn_readfirstlane(r.b.x);
    const uint32_t na = next_lane(r.a.x, b0);
    const uint32_t nb = next_lane(r.b.x, PADPAD);
    return match4(r.a, na, mk) || match4(r.b, nb, mk);
}

// V0: plain grid-stride uint4 stream (upper bound for this buffer)
__global__ __launch_bounds__(256) void k_stream(const uint4* p, size_t n, uint32_t mk, unsigned* hits) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, st = (size_t)gridDim.x * 256;
    int m = 0;
    for (; i + 3 * st < n; i += 4 * st) {
        uint4 a = p[i], b = p[i + st], c = p[i + 2 * st], d = p[i + 3 * st];
        m |= (a.x == mk) | (b.y == mk) | (c.z == mk) | (d.w == mk);
    }
    for (; i < n; i += st) { uint4 a = p[i]; m |= (a.x == mk); }
    if (m) atomicAdd(hits, 1u);
}

template <int DEPTH, int LDS_BYTES, int ATOM>
__global__ __launch_bounds__(256) void k_tiles(const uint16_t* tiles, const uint32_t* tile_len, uint32_t n_tiles, uint32_t mk, unsigned* hits) {
    __shared__ uint32_t lds[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
    if (LDS_BYTES > 0 && threadIdx.x == 0 && mk == 0x12345) lds[n_tiles % (LDS_BYTES / 4)] = 1;  // keep the allocation
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t stride = gridDim.x * 4;
    unsigned found = 0;
    for (uint32_t batch = blockIdx.x * 4 + wib; batch < n_tiles; batch += stride * 64u) {
        const unsigned long long my_tile = (unsigned long long)batch + (unsigned long long)lane * stride;
        const uint32_t my_len = my_tile < n_tiles ? tile_len[my_tile] : 0u;
        const uint32_t cnt = min(64u, (n_tiles - batch + stride - 1) / stride);
        TR q[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if ((uint32_t)d < cnt) q[d] = ld(tiles, batch + d * stride, __builtin_amdgcn_readlane(my_len, d), lane);
        for (uint32_t i = 0; i < cnt; i += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (i + d < cnt) {
                    const TR r = q[d];
                    const uint32_t len = __builtin_amdgcn_readlane(my_len, i + d);
                    if (i + d + DEPTH < cnt)
                        q[d] = ld(tiles, batch + (i + d + DEPTH) * stride, __builtin_amdgcn_readlane(my_len, i + d + DEPTH), lane);
                    if (len != 0 && __any(check(r, mk))) found++;
                }
            }
        }
    }
    if (found && lane == 0) atomicAdd(hits, found);
    if (ATOM == 1 && lane == 0) { atomicAdd(hits + 2, 1u); atomicAdd((unsigned long long*)(hits + 4), 1ull); }
    if (ATOM == 2 && threadIdx.x == 0) { atomicAdd(hits + 2, 1u); atomicAdd((unsigned long long*)(hits + 4), 1ull); }
    if (ATOM == 3 && threadIdx.x < 4) { atomicAdd((unsigned long long*)(hits + 4 + 2 * threadIdx.x), 1ull); }
    if (LDS_BYTES > 0 && mk == 0x12345) hits[1] = lds[0];
}

template <class F>
float time_it(F f, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    return best;
}

int main(int argc, char** argv) {
    uint32_t n_tiles = argc > 1 ? atoi(argv[1]) : 400000;  // 400k tiles = 800 MB
    uint32_t fill = argc > 2 ? atoi(argv[2]) : 1024;       // live slots per tile
    size_t n16 = (size_t)n_tiles * CAP;
    uint16_t* d_tiles; uint32_t* d_len; unsigned* d_hits;
    CK(hipMalloc(&d_tiles, n16 * 2)); CK(hipMalloc(&d_len, (size_t)n_tiles * 4)); CK(hipMalloc(&d_hits, 64));
    std::vector<uint16_t> h(n16);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < n16; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = (uint16_t)(x % 40000); }
    CK(hipMemcpy(d_tiles, h.data(), n16 * 2, hipMemcpyHostToDevice));
    std::vector<uint32_t> hl(n_tiles, fill);
    CK(hipMemcpy(d_len, hl.data(), (size_t)n_tiles * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_hits, 0, 64));
    const uint32_t mk = (50001u << 16) | 50000u;  // never matches
    double bytes = (double)n_tiles * fill * 2;
    printf("tiles %u, live slots/tile %u, %.1f MB per pass\n", n_tiles, fill, bytes / 1e6);
    for (int g : {1024, 2048, 4096, 8192}) {
        float ms = time_it([&] { hipLaunchKernelGGL(k_stream, dim3(g), dim3(256), 0, 0, (const uint4*)d_tiles, n16 / 8, mk, d_hits); });
        printf("stream      grid %5d : %7.3f ms  %7.1f GB/s\n", g, ms, (double)n16 * 2 / ms / 1e6);
    }
#define RUN(D, L, A) for (int g : {1280, 2560}) { \
        float ms = time_it([&] { hipLaunchKernelGGL((k_tiles<D, L, A>), dim3(g), dim3(256), 0, 0, d_tiles, d_len, n_tiles, mk, d_hits); }); \
        printf("tiles D=%d LDS=%5d ATOM=%d grid %5d : %7.3f ms  %7.1f GB/s\n", D, L, A, g, ms, bytes / ms / 1e6); }
    RUN(1, 25600, 0) RUN(1, 25600, 1) RUN(1, 25600, 2) RUN(1, 25600, 3) RUN(2, 25600, 0)
    unsigned hh[2]; CK(hipMemcpy(hh, d_hits, 8, hipMemcpyDeviceToHost));
    printf("hits %u\n", hh[0]);
    return 0;
}
