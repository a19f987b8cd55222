// tools/membench.hip -- text-streaming micro-benchmarks for choosing the scan kernels' load path.
// Not part of the product.  Build: hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o tools/membench
// Each kernel reads n bytes once and reduces them (so nothing is dead code); prints GB/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t mix(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// (a) fully coalesced: lane i reads 16 B at base + 16*i, 4 loads in flight
__global__ __launch_bounds__(1024) void k_coalesced(const uint4 *__restrict__ p, uint64_t n16, uint32_t *out)
{
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += mix(a) + mix(b) + mix(c) + mix(d);
    }
    for (; i < n16; i += stride) acc += mix(p[i]);
    if (acc == 0x12345678u) out[0] = acc;
}

// (b) segment per lane: lane owns SEGQ*16 contiguous bytes; NCH segments per lane per iteration, EXTRA halo pieces
template <int NCH, int EXTRA>
__global__ __launch_bounds__(1024) void k_segment(const uint8_t *__restrict__ text, uint64_t n, uint32_t *out)
{
    const uint64_t chunk = 64ull * 64 * NCH;
    const uint64_t nchunks = n / chunk - 1;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t k = wave; k < nchunks; k += nw) {
        uint4 v[NCH][4 + EXTRA];
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int q = 0; q < 4 + EXTRA; ++q)
                v[j][q] = *(const uint4 *)(text + k * chunk + ((uint64_t)j * 64 + lane) * 64 + 16 * q);
#pragma unroll
        for (int j = 0; j < NCH; ++j)
#pragma unroll
            for (int q = 0; q < 4 + EXTRA; ++q) acc += mix(v[j][q]);
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// (c) LDS staged: each wave copies 4 KiB (+64 B halo) coalesced into its private LDS slice with
// global_load_lds_dwordx4, then every lane reads its own 64 B (+16) with ds_read_b128
template <int DMA>
__global__ __launch_bounds__(1024) void k_lds_staged(const uint8_t *__restrict__ text, uint64_t n, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const uint64_t chunk = 4096;
    const uint64_t nchunks = n / chunk - 1;
    const uint32_t lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    unsigned char *mine = lds + wib * (4096 + 64);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t k = wave; k < nchunks; k += nw) {
        const uint8_t *src = text + k * chunk;
        if (DMA) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + q * 1024 + lane * 16),
                                                 (void __attribute__((address_space(3))) *)(mine + q * 1024), 16, 0, 0);
            if (lane < 4)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(src + 4096 + lane * 16),
                                                 (void __attribute__((address_space(3))) *)(mine + 4096), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            uint4 t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = *(const uint4 *)(src + q * 1024 + lane * 16);
            uint4 h = make_uint4(0, 0, 0, 0);
            if (lane < 4) h = *(const uint4 *)(src + 4096 + lane * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) *(uint4 *)(mine + q * 1024 + lane * 16) = t[q];
            if (lane < 4) *(uint4 *)(mine + 4096 + lane * 16) = h;
        }
        __builtin_amdgcn_wave_barrier();
        uint4 v[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) v[q] = *(const uint4 *)(mine + lane * 64 + 16 * q);
#pragma unroll
        for (int q = 0; q < 5; ++q) acc += mix(v[q]);
        __builtin_amdgcn_wave_barrier();
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename F> static void run(const char *name, uint64_t n, F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    std::vector<float> ms;
    for (int i = 0; i < 10; ++i) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    printf("%-44s median %.4f ms  %.1f GB/s   min %.4f ms %.1f GB/s\n", name, ms[5], n / ms[5] / 1e6, ms[0], n / ms[0] / 1e6);
}

int main(int argc, char **argv)
{
    uint64_t n = 1ull << 30;
    uint8_t *d; uint32_t *out;
    CK(hipMalloc(&d, n + 4096)); CK(hipMalloc(&out, 64));
    CK(hipMemset(d, 1, n + 4096));
    int cus = 256;
    for (int bpc = 1; bpc <= 2; ++bpc) {
        int grid = cus * bpc;
        printf("--- %d block(s) of 1024 threads per CU\n", bpc);
        run("coalesced 16B/lane x4", n, [&] { k_coalesced<<<grid, 1024>>>((const uint4 *)d, n / 16, out); });
        run("segment/lane NCH=1 +0", n, [&] { k_segment<1, 0><<<grid, 1024>>>(d, n, out); });
        run("segment/lane NCH=1 +1 halo", n, [&] { k_segment<1, 1><<<grid, 1024>>>(d, n, out); });
        run("segment/lane NCH=2 +1 halo", n, [&] { k_segment<2, 1><<<grid, 1024>>>(d, n, out); });
        run("segment/lane NCH=4 +1 halo", n, [&] { k_segment<4, 1><<<grid, 1024>>>(d, n, out); });
        int lds = 16 * (4096 + 64);
        run("LDS staged via registers", n, [&] { k_lds_staged<0><<<grid, 1024, lds>>>(d, n, out); });
        run("LDS staged via global_load_lds", n, [&] { k_lds_staged<1><<<grid, 1024, lds>>>(d, n, out); });
    }
    return 0;
}
