/*
 * csrc/ac_kernels.hip -- Aho-Corasick scan kernels for gfx950 (MI355X).
 *
 * ac_dfa_kernel    tuned path: complete DFA, hot rows staged in LDS once per
 *                  workgroup, 64-byte text segments streamed straight into
 *                  registers, NCH automata per lane, wave-level reduction and
 *                  one 64-bit atomic per wave.  Replaces ac_kernel3..5b
 *                  (cuda/cuda_ac.cu:23-532) -- no textures, no per-thread
 *                  counters copied back to the host (cuda/cuda_ac.cu:667-673).
 * ac_table_kernel  walks the reference-layout goto/supply/final tables from
 *                  HBM/L2 as given; replaces ac_kernel1/2 (cuda/cuda_ac.cu:535-592).
 *
 * Roofline: HBM read, 1 byte per text symbol.  No MFMA: the work
 * is one dependent table lookup per byte.
 */
#include <hip/hip_runtime.h>
#include "smh_launch.h"
#include "ac_lane.h"

#define SMH_AC_NCH 2

__device__ __forceinline__ void smh_wave_add(uint32_t cnt, uint64_t *count)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd((unsigned long long *)count, (unsigned long long)cnt);
}

template <typename E, int SIGMA, int HC, bool ALLHOT>
__global__ __launch_bounds__(SMH_BLOCK_THREADS) void ac_dfa_kernel(
    const uint8_t *__restrict__ text, uint64_t n, int m, const E *__restrict__ table, uint32_t hot_rows,
    uint32_t lds_bytes, int sigma_rt, const uint32_t *__restrict__ depth_first, uint64_t *count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    /* stage the hot rows: 16 bytes per lane per step, coalesced */
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(table);
        uint4 *dst = reinterpret_cast<uint4 *>(smh_lds);
        for (uint32_t i = threadIdx.x; i < lds_bytes / 16u; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const E *hot = reinterpret_cast<const E *>(smh_lds);
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t cnt = smh_ac_thread<E, SIGMA, HC, SMH_AC_NCH, ALLHOT>(gthread, nthreads, text, n, m, hot, table,
                                                                        hot_rows, sigma_rt, depth_first);
    smh_wave_add(cnt, count);
}

__global__ __launch_bounds__(256) void ac_table_kernel(const uint8_t *__restrict__ text, uint64_t n, int m,
                                                      const int32_t *__restrict__ transition,
                                                      const uint32_t *__restrict__ supply,
                                                      const uint32_t *__restrict__ final, int alphabet,
                                                      uint64_t *count)
{
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t cnt = smh_ac_table_thread(gthread, nthreads, text, n, m, transition, supply, final, alphabet);
    smh_wave_add(cnt, count);
}

/* ------------------------------------------------------------------ launch */
template <typename E, int SIGMA, int HC, bool ALLHOT>
static hipError_t launch_one(const smh_ac_launch &L, hipStream_t stream)
{
    auto kern = ac_dfa_kernel<E, SIGMA, HC, ALLHOT>;
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_bytes);
    if (err != hipSuccess) return err;
    int per_cu = 0;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, SMH_BLOCK_THREADS, L.lds_bytes);
    if (err != hipSuccess) return err;
    if (per_cu < 1) per_cu = 1;
    /* enough wave-chunks for every wave?  shrink the grid for small texts */
    const uint64_t chunk = (uint64_t)SMH_SEG * 64u * SMH_AC_NCH;
    const uint64_t n_chunks = (L.n + chunk - 1) / chunk;
    uint64_t blocks = (uint64_t)L.n_cus * (uint64_t)per_cu;
    const uint64_t want = (n_chunks + (SMH_BLOCK_THREADS / 64) - 1) / (SMH_BLOCK_THREADS / 64);
    if (blocks > want) blocks = want;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SMH_BLOCK_THREADS), L.lds_bytes, stream, L.d_text, L.n,
                       L.m, reinterpret_cast<const E *>(L.d_table), L.lds_rows, L.lds_bytes, L.alphabet,
                       L.d_depth_first, L.d_count);
    return hipGetLastError();
}

template <typename E, int SIGMA, int HC>
static hipError_t launch_hot(const smh_ac_launch &L, hipStream_t stream)
{
    return L.lds_rows >= L.rows ? launch_one<E, SIGMA, HC, true>(L, stream)
                                : launch_one<E, SIGMA, HC, false>(L, stream);
}

template <typename E, int SIGMA>
static hipError_t launch_halo(const smh_ac_launch &L, hipStream_t stream)
{
    const int halo = L.m - 1;
    if (halo <= 16) return launch_hot<E, SIGMA, 1>(L, stream);
    if (halo <= 32) return launch_hot<E, SIGMA, 2>(L, stream);
    if (halo <= 64) return launch_hot<E, SIGMA, 4>(L, stream);
    return launch_hot<E, SIGMA, 0>(L, stream);
}

hipError_t smh_launch_ac_dfa(const smh_ac_launch &L, hipStream_t stream)
{
    if (L.entry_bytes == 2)
        return L.alphabet == 4 ? launch_halo<uint16_t, 4>(L, stream) : launch_halo<uint16_t, 0>(L, stream);
    return L.alphabet == 4 ? launch_halo<uint32_t, 4>(L, stream) : launch_halo<uint32_t, 0>(L, stream);
}

hipError_t smh_launch_ac_table(const smh_ac_table_launch &L, hipStream_t stream)
{
    const uint64_t per_block = 256ull * SMH_AC_TABLE_SPAN;
    uint64_t blocks = (L.n + per_block - 1) / per_block;
    const uint64_t cap = (uint64_t)L.n_cus * 8u;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ac_table_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, L.d_text, L.n, L.m,
                       L.d_transition, L.d_supply, L.d_final, L.alphabet, L.d_count);
    return hipGetLastError();
}
