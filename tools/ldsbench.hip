// tools/ldsbench.hip -- LDS random-lookup throughput (dependent chains), to size the AC kernel.
// Not part of the product.  hipcc -O3 --offload-arch=gfx950 tools/ldsbench.hip -o tools/ldsbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

// MODE 0: ds_read_u16 of the entry; 1: ds_read_b32 of the containing dword + extract; 2: ds_read_b64 of the row + select;
// MODE 3: u32 entries, ds_read_b32
template <int MODE, int NCH, int STEPS>
__global__ __launch_bounds__(1024) void k_chain(const uint32_t *__restrict__ tab_g, uint32_t rows, uint32_t lds_bytes,
                                               const uint32_t *__restrict__ syms, uint32_t *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    for (uint32_t i = threadIdx.x; i < lds_bytes / 16; i += blockDim.x) ((uint4 *)lds)[i] = ((const uint4 *)tab_g)[i];
    __syncthreads();
    uint32_t row[NCH], acc = 0;
    uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < NCH; ++j) row[j] = (gid * 7 + j * 13) % rows;
    uint32_t w[NCH];
    for (int it = 0; it < STEPS / 16; ++it) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) w[j] = syms[(gid + it * 977 + j * 31) & 0xFFFFF];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const uint32_t c = (w[j] >> (2 * s)) & 3u;
                uint32_t e;
                if (MODE == 0) {
                    e = *(const uint16_t *)(lds + (row[j] * 4 + c) * 2);
                } else if (MODE == 1) {
                    const uint32_t d = *(const uint32_t *)(lds + row[j] * 8 + (c >> 1) * 4);
                    e = (d >> ((c & 1) * 16)) & 0xFFFFu;
                } else if (MODE == 2) {
                    const uint64_t d = *(const uint64_t *)(lds + row[j] * 8);
                    e = (uint32_t)(d >> (c * 16)) & 0xFFFFu;
                } else {
                    e = *(const uint32_t *)(lds + (row[j] * 4 + c) * 4);
                }
                acc += e >> 15;
                row[j] = e & 0x7FFFu;
            }
        }
    }
    uint32_t r = acc;
#pragma unroll
    for (int j = 0; j < NCH; ++j) r += row[j];
    if (r == 0x12345678u) out[0] = r;
}

template <int MODE, int NCH>
static void bench(const char *name, uint32_t rows, int bpc, const uint32_t *d_syms, uint32_t *d_out, bool skew)
{
    const int STEPS = 4096;
    std::vector<uint32_t> host;
    uint32_t lds_bytes;
    // transitions: skewed like a real AC DFA (most targets among the first rows/16 rows) or uniform
    auto target = [&](uint32_t i) { uint32_t r = (uint32_t)rand(); return skew ? ((r & 7) ? (r >> 3) % (rows / 16 + 1) : (r >> 3) % rows) : r % rows; };
    if (MODE == 3) {
        host.resize(rows * 4);
        for (auto &e : host) e = target(0) | ((rand() & 63) == 0 ? 0x8000u : 0);
        lds_bytes = rows * 16;
    } else {
        std::vector<uint16_t> h16(rows * 4);
        for (auto &e : h16) e = (uint16_t)(target(0) | ((rand() & 63) == 0 ? 0x8000u : 0));
        host.resize(rows * 2);
        memcpy(host.data(), h16.data(), rows * 8);
        lds_bytes = rows * 8;
    }
    uint32_t *d_tab;
    CK(hipMalloc(&d_tab, lds_bytes + 64));
    CK(hipMemcpy(d_tab, host.data(), lds_bytes, hipMemcpyHostToDevice));
    auto kern = k_chain<MODE, NCH, STEPS>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    int grid = 256 * bpc;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int i = 0; i < 7; ++i) {
        CK(hipEventRecord(a));
        kern<<<grid, 1024, lds_bytes>>>(d_tab, rows, lds_bytes, d_syms, d_out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b)); ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    double lookups = (double)grid * 1024 * NCH * STEPS;
    printf("%-34s rows %6u lds %6u KB %s bpc %d NCH %d: %.3f ms  %.2f T lookups/s  = %.2f per clk per CU @2.4GHz\n", name, rows,
           lds_bytes >> 10, skew ? "skew" : "unif", bpc, NCH, ms[3], lookups / ms[3] / 1e9, lookups / ms[3] / 1e9 * 1e12 / 256 / 2.4e9 / 1e3 * 1e3 / 1e3);
    CK(hipFree(d_tab));
}

int main()
{
    uint32_t *d_syms, *d_out;
    std::vector<uint32_t> syms(1 << 20);
    for (auto &s : syms) s = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
    CK(hipMalloc(&d_syms, syms.size() * 4)); CK(hipMalloc(&d_out, 64));
    CK(hipMemcpy(d_syms, syms.data(), syms.size() * 4, hipMemcpyHostToDevice));
    for (int skew = 0; skew < 2; ++skew) {
        for (uint32_t rows : {2812u, 10816u}) {
            for (int bpc = 1; bpc <= (rows < 9000 ? 2 : 1); ++bpc) {
                bench<0, 1>("u16 entry ds_read_u16", rows, bpc, d_syms, d_out, skew);
                bench<0, 2>("u16 entry ds_read_u16", rows, bpc, d_syms, d_out, skew);
                bench<0, 4>("u16 entry ds_read_u16", rows, bpc, d_syms, d_out, skew);
                bench<1, 2>("u16 entry via ds_read_b32", rows, bpc, d_syms, d_out, skew);
                bench<2, 2>("u16 row via ds_read_b64", rows, bpc, d_syms, d_out, skew);
                if (rows * 16 <= 160 * 1024) bench<3, 2>("u32 entry ds_read_b32", rows, bpc, d_syms, d_out, skew);
            }
        }
    }
    return 0;
}
