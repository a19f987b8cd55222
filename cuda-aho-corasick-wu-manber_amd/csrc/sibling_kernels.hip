/*
 * csrc/sibling_kernels.hip -- table-walking kernels of the sibling algorithms (Set-Horspool, SBOM, SOG) for gfx950.
 *
 * sh_table_kernel  the reference-layout reversed trie (state_transition / state_final as
 *                  preproc_sh fills them) walked from HBM/L2 as given, the caller's bmBc staged in
 *                  LDS and driving a per-lane skip loop; replaces sh_kernel1..5
 *                  (cuda/cuda_sh.cu:23-108 and its four siblings).  Latency bound (dependent L2
 *                  lookups); the tuned Set-Horspool path runs the Wu-Manber / automaton kernels
 *                  (sh_host.c).
 */
#include <hip/hip_runtime.h>
#include "smh_launch.h"
#include "sh_lane.h"
#include "sbom_lane.h"
#include "sog_lane.h"

__global__ __launch_bounds__(256) void sh_table_kernel(const uint8_t *__restrict__ text, uint64_t n, int m, int alphabet,
                                                      const int32_t *__restrict__ transition,
                                                      const uint32_t *__restrict__ final_,
                                                      const int32_t *__restrict__ bmbc_g, uint64_t *count)
{
    __shared__ int32_t bmbc[256];
    for (int i = threadIdx.x; i < alphabet; i += blockDim.x) bmbc[i] = bmbc_g[i];
    __syncthreads();
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    uint32_t cnt = smh_sh_table_thread<int32_t>(gthread, nthreads, text, n, transition, final_, bmbc, m, alphabet);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd((unsigned long long *)count, (unsigned long long)cnt);
}

hipError_t smh_launch_sh_table(const smh_sh_table_launch &L, hipStream_t stream)
{
    const uint64_t per_block = 256ull * SMH_SH_TABLE_SPAN;
    uint64_t blocks = (L.n + per_block - 1) / per_block;
    const uint64_t cap = (uint64_t)L.n_cus * 8u;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sh_table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, L.d_text, L.n, L.m, L.alphabet,
                       L.d_transition, L.d_final, L.d_bmbc, L.d_count);
    return hipGetLastError();
}

/* sbom_table_kernel  the factor oracle and its per-state pattern lists (packed: offsets + ids) walked
 *                    from HBM/L2 as given; replaces sbom_kernel1..5 (cuda/cuda_sbom.cu:23-123 and
 *                    siblings).  Latency bound. */
__global__ __launch_bounds__(256) void sbom_table_kernel(const uint8_t *__restrict__ text, uint64_t n, int m, int alphabet,
                                                        const int32_t *__restrict__ transition,
                                                        const uint32_t *__restrict__ final_off,
                                                        const uint32_t *__restrict__ final_ids,
                                                        const uint8_t *__restrict__ patterns, uint64_t *count)
{
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    uint32_t cnt = smh_sbom_table_thread(gthread, nthreads, text, n, transition, final_off, final_ids, patterns, m, alphabet);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd((unsigned long long *)count, (unsigned long long)cnt);
}

hipError_t smh_launch_sbom_table(const smh_sbom_table_launch &L, hipStream_t stream)
{
    const uint64_t per_block = 256ull * SMH_SBOM_TABLE_SPAN;
    uint64_t blocks = (L.n + per_block - 1) / per_block;
    const uint64_t cap = (uint64_t)L.n_cus * 8u;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sbom_table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, L.d_text, L.n, L.m, L.alphabet,
                       L.d_transition, L.d_final_off, L.d_final_ids, L.d_patterns, L.d_count);
    return hipGetLastError();
}

/* sog_table_kernel  the caller's SOG tables walked as given: T8 (16 MiB) from HBM/L2 with the shift-or state per
 *                   lane, then the 2-level bitmap, the binary search and the compare; replaces sog_kernel1..5
 *                   (cuda/cuda_sog.cu:60-218 and siblings).  Latency bound (one dependent T8 read per column). */
__global__ __launch_bounds__(256) void sog_table_kernel(const uint8_t *__restrict__ text, uint64_t n, const uint8_t *__restrict__ t8,
                                                       const uint32_t *__restrict__ hs_sorted, const int32_t *__restrict__ index,
                                                       const uint8_t *__restrict__ hs2, const uint8_t *__restrict__ patterns,
                                                       int p_size, uint64_t *count)
{
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    uint32_t cnt = smh_sog_table_thread(gthread, nthreads, text, n, t8, hs_sorted, index, hs2, patterns, p_size);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd((unsigned long long *)count, (unsigned long long)cnt);
}

hipError_t smh_launch_sog_table(const smh_sog_table_launch &L, hipStream_t stream)
{
    const uint64_t per_block = 256ull * SMH_SOG_TABLE_SPAN;
    uint64_t blocks = (L.n + per_block - 1) / per_block;
    const uint64_t cap = (uint64_t)L.n_cus * 8u;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sog_table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, L.d_text, L.n, L.d_t8, L.d_hs, L.d_index,
                       L.d_hs2, L.d_patterns, L.p_size, L.d_count);
    return hipGetLastError();
}
