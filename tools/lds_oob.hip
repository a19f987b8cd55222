// Development probe (hipcc -O2 --offload-arch=gfx950 -o tools/lds_oob tools/lds_oob.hip): what does a ds_read beyond the workgroup's LDS allocation return on gfx950?
// (the hybrid AC image lets lanes that sit in a compact row issue the full-row lookup with their raw id)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
extern __shared__ unsigned char lds[];
__global__ void probe(uint32_t *out, uint32_t off)
{
    for (uint32_t i = threadIdx.x; i < 40000u; i += blockDim.x) reinterpret_cast<uint32_t *>(lds)[i] = 0xABCD0000u + i;
    __syncthreads();
    const uint32_t addr = off + 2u * threadIdx.x;
    const uint16_t v = *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>(addr);
    const uint16_t w = *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>(2u * threadIdx.x);
    out[threadIdx.x] = ((uint32_t)v << 16) | w;
}
int main()
{
    uint32_t *d, h[256];
    hipMalloc(&d, sizeof h);
    hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    const uint32_t offs[] = {1000u, 160000u, 0x28000u, 0x40000u, 0x100000u, 0x1F0000u};
    for (uint32_t off : offs) {
        hipMemset(d, 0xFF, sizeof h);
        hipLaunchKernelGGL(probe, dim3(1), dim3(256), 160000, 0, d, off);
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("off 0x%06x: err %d  lane0 %08x lane1 %08x lane255 %08x\n", off, (int)e, h[0], h[1], h[255]);
        if (e != hipSuccess) return 1;
    }
    return 0;
}
