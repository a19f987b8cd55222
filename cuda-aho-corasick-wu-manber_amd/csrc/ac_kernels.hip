/*
 * csrc/ac_kernels.hip -- Aho-Corasick scan kernels for gfx950 (MI355X).
 *
 * ac_dfa_kernel    tuned path: the depth-K automaton staged in LDS once per
 *                  workgroup (stride 1 or 2 symbols per lookup), 64-byte text
 *                  segments streamed straight into registers, NCH automata per
 *                  lane, candidates compacted into a per-wave queue (ballot +
 *                  prefix count) and verified against the full DFA in HBM,
 *                  wave-level reduction and one 64-bit atomic per wave.
 *                  Replaces ac_kernel3..5b (cuda/cuda_ac.cu:23-532) -- no
 *                  textures, no per-thread counters copied back to the host
 *                  (cuda/cuda_ac.cu:667-673).
 * ac_table_kernel  walks the reference-layout goto/supply/final tables from
 *                  HBM/L2 as given; replaces ac_kernel1/2 (cuda/cuda_ac.cu:535-592).
 *
 * Roofline: HBM read, 1 byte per text symbol.  No MFMA: the work
 * is one dependent table lookup per byte.
 */
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "smh_launch.h"
#include "ac_lane.h"

#define SMH_AC_NCH 1 /* text segments (automata) per lane; measured best with the prefetch: profiles/ */

__device__ __forceinline__ void smh_wave_add(uint32_t cnt, uint64_t *count)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd((unsigned long long *)count, (unsigned long long)cnt);
}

/* Workgroup-level count: wave sums meet in LDS and ONE 64-bit atomic per workgroup reaches HBM.
 * With one atomic per wave, 4096 same-address atomics (~12 ns each, serialised at the memory side)
 * queued up when the balanced waves all finish together: a ~45 us tail on a 240 us kernel whenever
 * every wave has matches.  `lds` may be the table region: the first barrier makes sure every wave
 * is done reading it. */
__device__ __forceinline__ void smh_block_add(uint32_t cnt, uint64_t *count, unsigned char *lds)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    __syncthreads();
    uint32_t *part = reinterpret_cast<uint32_t *>(lds);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x < 64) {
        uint64_t v = threadIdx.x < (blockDim.x >> 6) ? part[threadIdx.x] : 0u;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0 && v) atomicAdd((unsigned long long *)count, (unsigned long long)v);
    }
}

template <typename E, int SIGMA, int STRIDE, int HC, bool EXACT, int NCH = SMH_AC_NCH, bool PF = true, int SW = 16, bool POS = false>
__global__ __launch_bounds__(SMH_BLOCK_THREADS) void ac_dfa_kernel(const E *__restrict__ scan_table, uint32_t lds_bytes,
                                                                  smh_ac_verify_ctx V, smh_ac_df df,
                                                                  uint64_t *queue_base, uint64_t *count,
                                                                  uint32_t full_rows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    /* stage the depth-K automaton: 16 bytes per lane per step, coalesced; four loads in flight per
     * lane so that the staging costs about one memory round trip per 64 KiB, not one per 16 KiB */
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(scan_table);
        uint4 *dst = reinterpret_cast<uint4 *>(smh_lds);
        const uint32_t n16 = lds_bytes / 16u;
        uint32_t i = threadIdx.x;
        for (; i + 3u * SMH_BLOCK_THREADS < n16; i += 4u * SMH_BLOCK_THREADS) {
            const uint4 t0 = src[i], t1 = src[i + SMH_BLOCK_THREADS], t2 = src[i + 2u * SMH_BLOCK_THREADS],
                        t3 = src[i + 3u * SMH_BLOCK_THREADS];
            dst[i] = t0;
            dst[i + SMH_BLOCK_THREADS] = t1;
            dst[i + 2u * SMH_BLOCK_THREADS] = t2;
            dst[i + 3u * SMH_BLOCK_THREADS] = t3;
        }
        for (; i < n16; i += SMH_BLOCK_THREADS) dst[i] = src[i];
    }
    __syncthreads();
    /* the lane code addresses the table by LDS byte offset: it must sit at offset 0 (dynamic LDS only) */
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    uint32_t cnt;
    if constexpr (STRIDE == 3) { /* hybrid stride 2: full rows + compact item lists */
        const smh_fmt_s2h fmt{full_rows, full_rows * 28u};
        cnt = smh_ac_thread<smh_fmt_s2h, HC, NCH, EXACT, PF, SW, POS>(fmt, gthread, nthreads, smh_lds, V, df, queue_base);
    } else if constexpr (STRIDE == 2) {
        cnt = smh_ac_thread<smh_fmt_s2, HC, NCH, EXACT, PF, SW, POS>(smh_fmt_s2{}, gthread, nthreads, smh_lds, V, df, queue_base);
    } else {
        const smh_fmt_s1<E, SIGMA> fmt{V.sigma};
        cnt = smh_ac_thread<smh_fmt_s1<E, SIGMA>, HC, NCH, EXACT, PF, SW, POS>(fmt, gthread, nthreads, smh_lds, V, df, queue_base);
    }
    if constexpr (!POS) smh_block_add(cnt, count, smh_lds); /* positions mode: the cursor is the count */
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

__global__ __launch_bounds__(256) void ac_positions_kernel(smh_ac_verify_ctx V, uint64_t *positions, uint64_t capacity,
                                                          uint64_t *cursor)
{
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    smh_ac_positions_thread(gthread, nthreads, V, positions, capacity, cursor);
}

hipError_t smh_launch_ac_positions(const smh_ac_verify_ctx &V, uint64_t *d_positions, uint64_t capacity,
                                   uint64_t *d_cursor, int n_cus, hipStream_t stream)
{
    const uint64_t per_block = 256ull * SMH_SEG;
    uint64_t blocks = (V.n + per_block - 1) / per_block;
    const uint64_t cap = (uint64_t)n_cus * 8u;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ac_positions_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, V, d_positions, capacity,
                       d_cursor);
    return hipGetLastError();
}

/* ------------------------------------------------------------------ launch */
uint32_t smh_ac_max_blocks(int n_cus) { return (uint32_t)n_cus * 2u; }

/* development knob (not part of the API): SMH_AC_TUNE="bpc=1|2" caps the workgroups per CU.  Other
 * variants that were measured and dropped (two or four segments per lane, no software prefetch,
 * 128-byte segments) remain available as template parameters: profiles/README.md */
static int tune_get(const char *key, int dflt)
{
    const char *t = getenv("SMH_AC_TUNE");
    if (!t) return dflt;
    const char *p = strstr(t, key);
    if (!p) return dflt;
    return atoi(p + strlen(key) + 1);
}

template <typename E, int SIGMA, int STRIDE, int HC, bool EXACT, bool POS = false>
static hipError_t launch_one(const smh_ac_launch &L, hipStream_t stream)
{
    constexpr int NCH = SMH_AC_NCH, SW = 16;
    auto kern = ac_dfa_kernel<E, SIGMA, STRIDE, HC, EXACT, NCH, true, SW, POS>;
    /* the attribute call and the occupancy query cost tens of microseconds of host time, during
     * which the GPU idles between the caller's events: do them once per (kernel, LDS size) */
    static uint32_t cached_lds = 0xFFFFFFFFu;
    static int cached_per_cu = 0;
    if (cached_lds != L.lds_bytes) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.lds_bytes);
        if (err != hipSuccess) return err;
        int q = 0;
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, SMH_BLOCK_THREADS, L.lds_bytes);
        if (err != hipSuccess) return err;
        cached_per_cu = q;
        cached_lds = L.lds_bytes;
    }
    int per_cu = cached_per_cu;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    if (per_cu > tune_get("bpc", 2)) per_cu = tune_get("bpc", 2);
    /* enough wave-chunks for every wave?  shrink the grid for small texts */
    const uint64_t chunk = (uint64_t)(4u * SW) * 64u * NCH;
    const uint64_t n_chunks = (L.V.n + chunk - 1) / chunk;
    uint64_t blocks = (uint64_t)L.n_cus * (uint64_t)per_cu;
    const uint64_t want = (n_chunks + (SMH_BLOCK_THREADS / 64) - 1) / (SMH_BLOCK_THREADS / 64);
    if (blocks > want) blocks = want;
    if (blocks < 1) blocks = 1;
    if (blocks > smh_ac_max_blocks(L.n_cus)) blocks = smh_ac_max_blocks(L.n_cus); /* the queue workspace is sized for this */
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SMH_BLOCK_THREADS), L.lds_bytes, stream,
                       reinterpret_cast<const E *>(L.d_scan_table), L.lds_bytes, L.V, L.df, L.d_queue, L.d_count,
                       L.full_rows);
    return hipGetLastError();
}

template <typename E, int SIGMA, int STRIDE, int HC, bool POS>
static hipError_t launch_exact(const smh_ac_launch &L, hipStream_t stream)
{
    if constexpr (POS && HC > 2) {
        return hipErrorNotSupported; /* match recording covers a 32-byte halo: the caller falls back */
    } else if constexpr (STRIDE == 3) {
        if (L.exact) return launch_one<E, SIGMA, 3, HC, true, POS>(L, stream);
        if constexpr (HC <= 2) return launch_one<E, SIGMA, 3, HC, false, POS>(L, stream);
        return hipErrorInvalidValue; /* the plan keeps K - 1 <= 32 for a depth-cut hybrid image */
    } else {
        return L.exact ? launch_one<E, SIGMA, STRIDE, HC, true, POS>(L, stream)
                       : launch_one<E, SIGMA, STRIDE, HC, false, POS>(L, stream);
    }
}

template <typename E, int SIGMA, int STRIDE, bool POS>
static hipError_t launch_halo(const smh_ac_launch &L, hipStream_t stream)
{
    const int halo = L.V.K - 1;
    if (halo <= 16) return launch_exact<E, SIGMA, STRIDE, 1, POS>(L, stream);
    if (halo <= 32) return launch_exact<E, SIGMA, STRIDE, 2, POS>(L, stream);
    return launch_exact<E, SIGMA, STRIDE, 4, POS>(L, stream);
}

template <bool POS>
static hipError_t launch_dfa(const smh_ac_launch &L, hipStream_t stream)
{
    if (L.V.K - 1 > 64) return hipErrorInvalidValue;
    if (L.stride == 2) {
        if (L.V.sigma != 4 || L.scan_entry_bytes != 2) return hipErrorInvalidValue;
        if (L.full_rows) return launch_halo<uint16_t, 4, 3, POS>(L, stream);
        return launch_halo<uint16_t, 4, 2, POS>(L, stream);
    }
    if (L.scan_entry_bytes == 2)
        return L.V.sigma == 4 ? launch_halo<uint16_t, 4, 1, POS>(L, stream) : launch_halo<uint16_t, 0, 1, POS>(L, stream);
    return L.V.sigma == 4 ? launch_halo<uint32_t, 4, 1, POS>(L, stream) : launch_halo<uint32_t, 0, 1, POS>(L, stream);
}

hipError_t smh_launch_ac_dfa(const smh_ac_launch &L, hipStream_t stream) { return launch_dfa<false>(L, stream); }

/* positions mode of the same kernels (L.po set); hipErrorNotSupported when the plan's halo exceeds 32
 * bytes -- the caller then runs ac_positions_kernel */
hipError_t smh_launch_ac_dfa_positions(const smh_ac_launch &L, hipStream_t stream) { return launch_dfa<true>(L, stream); }

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
