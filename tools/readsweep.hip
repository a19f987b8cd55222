// tools/readsweep.hip -- which read-only kernel streams 1 GiB fastest on this device?  (round 3: the bench's
// "stream_read" probe was beaten by real scan kernels; this sweep picks the probe's shape.)  Not part of the product.
// Build: hipcc -O3 --offload-arch=gfx950 tools/readsweep.hip -o tools/readsweep
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// grid-stride: lane i reads 16 B at 16*i; U loads in flight per lane
template <int U>
__global__ __launch_bounds__(1024) void k_grid(const v4u *__restrict__ p, uint64_t n16, unsigned long long *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i + (uint64_t)(U - 1) * stride < n16; i += (uint64_t)U * stride) {
        v4u v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = p[i + (uint64_t)q * stride];
#pragma unroll
        for (int q = 0; q < U; ++q) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
    }
    for (; i < n16; i += stride) { v4u a = p[i]; acc ^= a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) atomicXor(out, (unsigned long long)acc);
}

// block-contiguous: workgroup w streams its own contiguous slab, wave-coalesced 1 KiB per load instruction, U in flight
template <int U>
__global__ __launch_bounds__(1024) void k_slab(const v4u *__restrict__ p, uint64_t n16, unsigned long long *out)
{
    const uint64_t per = n16 / gridDim.x;
    const v4u *b = p + (uint64_t)blockIdx.x * per;
    uint32_t acc = 0;
    uint64_t i = threadIdx.x;
    for (; i + (uint64_t)(U - 1) * blockDim.x < per; i += (uint64_t)U * blockDim.x) {
        v4u v[U];
#pragma unroll
        for (int q = 0; q < U; ++q) v[q] = b[i + (uint64_t)q * blockDim.x];
#pragma unroll
        for (int q = 0; q < U; ++q) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
    }
    if (acc == 0x12345678u) atomicXor(out, (unsigned long long)acc);
}

// scan-shaped: 4 KiB wave-chunks, lane owns 64 B (four 16 B loads), chunks dealt round-robin to waves; C chunks in flight per wave
template <int C>
__global__ __launch_bounds__(1024) void k_chunk(const uint8_t *__restrict__ t, uint64_t n_chunks, unsigned long long *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    for (uint64_t k = wave; k + (uint64_t)(C - 1) * nw < n_chunks; k += (uint64_t)C * nw) {
        v4u v[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[c][q] = *(const v4u *)(t + (k + (uint64_t)c * nw) * 4096u + lane * 64u + 16u * q);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc ^= v[c][q].x ^ v[c][q].y ^ v[c][q].z ^ v[c][q].w;
    }
    if (acc == 0x12345678u) atomicXor(out, (unsigned long long)acc);
}

template <typename F> static double timeit(F launch)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int i = 0; i < 9; ++i) {
        CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    return ts[4];
}

int main()
{
    const uint64_t n = 1ull << 30;
    uint8_t *d; unsigned long long *out;
    CK(hipMalloc(&d, n + 4096)); CK(hipMalloc(&out, 8)); CK(hipMemset(d, 1, n + 4096)); CK(hipMemset(out, 0, 8));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    printf("%s, %d CUs\n", pr.name, cus);
    const int threads[] = {256, 512, 1024};
    const int per_cu[] = {1, 2, 4};
#define RUN(NAME, KERN, ARGS_N)                                                                                   \
    for (int th : threads) for (int pc : per_cu) {                                                                \
        if (th * pc > 2048) continue;                                                                             \
        double ms = timeit([&] { hipLaunchKernelGGL(KERN, dim3(cus * pc), dim3(th), 0, 0, ARGS_N); });            \
        printf("%-28s threads %4d x %d/CU (%2d waves/CU): %.4f ms  %.0f GB/s\n", NAME, th, pc, th * pc / 64, ms, n / ms / 1e6); \
    }
#define A16 (const v4u *)d, n / 16, out
#define ACH (const uint8_t *)d, n / 4096, out
    RUN("grid-stride U=4", k_grid<4>, A16)
    RUN("grid-stride U=8", k_grid<8>, A16)
    RUN("grid-stride U=16", k_grid<16>, A16)
    RUN("slab U=4", k_slab<4>, A16)
    RUN("slab U=8", k_slab<8>, A16)
    RUN("slab U=16", k_slab<16>, A16)
    RUN("4KiB chunks C=1", k_chunk<1>, ACH)
    RUN("4KiB chunks C=2", k_chunk<2>, ACH)
    RUN("4KiB chunks C=4", k_chunk<4>, ACH)
    return 0;
}
