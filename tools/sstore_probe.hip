// tools/sstore_probe.hip -- does this device execute scalar stores (s_store_dwordx2 + s_dcache_wb)?  One wave, 3 entries.
// Round 3: the candidate queue of the depth-cut automaton kernels is written with scalar stores so that the scan
// kernels contain no VMEM store (a vector store anywhere in the kernel cost the scan 5-7 %: csrc/ac_lane.h smh_ac_emit).
// Build: hipcc -O3 --offload-arch=gfx950 tools/sstore_probe.hip -o tools/sstore_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long *q, const unsigned *in, unsigned *out)
{
    const unsigned v = in[threadIdx.x];
    unsigned long long mask = __ballot(v > 100);
    unsigned cnt = 0;
    while (mask) {
        const int l = __builtin_ctzll(mask);
        mask &= mask - 1;
        const unsigned lo = __builtin_amdgcn_readlane(v, l);
        const unsigned long long ent = ((unsigned long long)l << 32) | lo;
        asm volatile("s_store_dwordx2 %0, %1, %2" ::"s"(ent), "s"(q), "s"(cnt * 8u) : "memory");
        cnt++;
    }
    asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
    out[threadIdx.x] = (unsigned)__hip_atomic_load(q + (threadIdx.x % 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + cnt;
}
int main()
{
    unsigned h[64], *din, *dout;
    unsigned long long *q, hq[4] = {0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) h[i] = i == 5 ? 500 : i == 17 ? 700 : i == 63 ? 900 : i;
    hipMalloc(&din, 256); hipMalloc(&dout, 256); hipMalloc(&q, 64);
    hipMemset(q, 0, 64);
    hipMemcpy(din, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, q, din, dout);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(hq, q, 32, hipMemcpyDeviceToHost);
    hipMemcpy(h, dout, 256, hipMemcpyDeviceToHost);
    printf("sync %s; queue %llx %llx %llx; out[0..2] %u %u %u\n", hipGetErrorString(e), hq[0], hq[1], hq[2], h[0], h[1], h[2]);
    const bool ok = e == hipSuccess && hq[0] == ((5ull << 32) | 500) && hq[1] == ((17ull << 32) | 700) && hq[2] == ((63ull << 32) | 900) &&
                    h[0] == 503 && h[1] == 703 && h[2] == 903;
    printf(ok ? "SCALAR STORES OK\n" : "SCALAR STORES NOT OK\n");
    return ok ? 0 : 1;
}
