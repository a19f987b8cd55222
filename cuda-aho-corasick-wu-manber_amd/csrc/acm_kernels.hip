/*
 * csrc/acm_kernels.hip -- the mixed-length automaton kernel for gfx950 (acm_host.c, acm_lane.h).
 *
 * One pass over the text for a set of patterns of different lengths: the depth-K automaton with joined output
 * counts staged in LDS once per workgroup, 64-byte text segments in registers, end ownership with a K-1 byte
 * warm-up, candidates for the patterns longer than K compacted per wave (ballot + prefix count) and walked down
 * the goto trie in HBM.  Replaces nothing in the reference, whose automaton cannot count such a set
 * (ac/ac.c:118 "Join outputs missing"); the expected value is the length-class decomposition.
 * Roofline: HBM read, 1 byte per text symbol; one LDS lookup per byte.
 */
#include <hip/hip_runtime.h>
#include "smh_launch.h"
#include "acm_lane.h"

template <typename E, int SIGMA>
__global__ __launch_bounds__(SMH_BLOCK_THREADS) void acm_kernel(const E *__restrict__ scan_g, uint32_t lds_bytes, smh_acm_ctx C,
                                                               uint64_t *queue_base, uint64_t *count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(scan_g);
        uint4 *dst = reinterpret_cast<uint4 *>(smh_lds);
        for (uint32_t i = threadIdx.x; i < lds_bytes / 16u; i += SMH_BLOCK_THREADS) dst[i] = src[i];
    }
    const smh_chunk_sched S = smh_sched_init(smh_lds, lds_bytes);
    __syncthreads();
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = smh_acm_thread<E, SIGMA>(gthread, S, smh_lds, scan_g, C, queue_base);
    /* wave sums meet in LDS, one 64-bit atomic per workgroup (ac_kernels.inc smh_block_add) */
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    __syncthreads();
    uint32_t *part = reinterpret_cast<uint32_t *>(smh_lds);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x < 64) {
        uint64_t v = threadIdx.x < (blockDim.x >> 6) ? part[threadIdx.x] : 0u;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (threadIdx.x == 0 && v) atomicAdd((unsigned long long *)count, (unsigned long long)v);
    }
}

uint32_t smh_acm_max_blocks(int n_cus) { return (uint32_t)n_cus * 2u; }

template <typename E, int SIGMA>
static hipError_t launch(const smh_acm_launch &L, hipStream_t stream)
{
    auto kern = acm_kernel<E, SIGMA>;
    static smh_attr_cache cache;
    int per_cu = 0;
    const hipError_t err = cache.get(kern, L.lds_bytes + SMH_SCHED_LDS, SMH_BLOCK_THREADS, &per_cu);
    if (err != hipSuccess) return err;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    const uint64_t chunk = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (L.C.n + chunk - 1) / chunk;
    uint64_t blocks = (uint64_t)L.n_cus * (uint64_t)per_cu;
    const uint64_t want = (n_chunks + (SMH_BLOCK_THREADS / 64) - 1) / (SMH_BLOCK_THREADS / 64);
    if (blocks > want) blocks = want;
    if (blocks < 1) blocks = 1;
    if (blocks > smh_acm_max_blocks(L.n_cus)) blocks = smh_acm_max_blocks(L.n_cus); /* the queue workspace is sized for this */
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SMH_BLOCK_THREADS), L.lds_bytes + SMH_SCHED_LDS, stream,
                       reinterpret_cast<const E *>(L.d_scan), L.lds_bytes, L.C, L.d_queue, L.d_count);
    return hipGetLastError();
}

hipError_t smh_launch_acm(const smh_acm_launch &L, hipStream_t stream)
{
    if (L.entry_bytes == 2) return L.C.sigma == 4 ? launch<uint16_t, 4>(L, stream) : launch<uint16_t, 0>(L, stream);
    return L.C.sigma == 4 ? launch<uint32_t, 4>(L, stream) : launch<uint32_t, 0>(L, stream);
}
