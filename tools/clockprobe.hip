// tools/clockprobe.hip -- shader clock held under load, to turn kernel times into cycles.
// Not part of the product.  hipcc -O3 --offload-arch=gfx950 tools/clockprobe.hip -o tools/clockprobe
//
// Every CU runs 16 waves; each wave alternates dependent VALU ops with LDS lookups (MODE 1) or runs VALU
// only (MODE 0).  s_memtime counts shader-clock cycles, s_memrealtime the constant 100 MHz reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k_probe(uint64_t *out, uint32_t iters, uint32_t seed)
{
    __shared__ uint32_t lds[16384];
    for (uint32_t i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (i * 2654435761u + seed) & 16383u;
    __syncthreads();
    const uint64_t c0 = __builtin_readcyclecounter();      // s_memtime
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    uint32_t x = threadIdx.x + seed, acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        if (MODE == 1) x = lds[(x + acc) & 16383u];
        x = x * 5u + 1u;
        acc += x >> 3;
        x ^= acc;
    }
    const uint64_t c1 = __builtin_readcyclecounter();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    if (acc == 0x12345678u) out[3] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}

int main()
{
    uint64_t *d, h[4];
    CK(hipMalloc(&d, 64));
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(d, 0, 64));
            if (mode == 0) k_probe<0><<<256, 1024>>>(d, 200000, rep);
            else k_probe<1><<<256, 1024>>>(d, 100000, rep);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, d, 32, hipMemcpyDeviceToHost));
            printf("mode %d (%s): %llu shader cycles in %llu ticks of 100 MHz -> %.3f GHz\n", mode,
                   mode ? "VALU + LDS lookups" : "VALU only", (unsigned long long)h[0], (unsigned long long)h[1],
                   (double)h[0] / ((double)h[1] * 10.0));
        }
    return 0;
}
