/*
 * csrc/hash_kernels.hip -- the window-hash engine's kernel for gfx950 (hash_engine.h, hash_host.c, hash_lane.h).
 *
 * The Bloom filter of the patterns' rolling window hashes (up to 128 KiB) is staged in LDS once per 1024-thread workgroup, the
 * waves' candidate queues sit behind it; the text streams through registers in 4 KiB wave-chunks taken from the workgroup's LDS
 * counter (lane_common.h).  Stage 1 costs every column eleven VALU and one ds_read_b32 whatever the text; stage 2 -- the window
 * from L2, both cuckoo slots of the pattern table from L2 / Infinity Cache, compared in registers -- two dependent round trips per
 * surviving column.  Replaces wm_kernel* (cuda/cuda_wm.cu:60-650) for large byte sets on text that defeats q-gram filters.
 * Roofline: HBM read, 1 byte per text symbol; bound by VALU issue and the LDS lookup rate (stage 1), then by L2 latency (stage 2).
 */
#include <hip/hip_runtime.h>
#include "smh_launch.h"
#include "hash_lane.h"
#include "smh_stats.h"

#define SMH_HASH_QUEUES ((SMH_BLOCK_THREADS / 64) * SMH_HASH_QCAP * 4u)

template <bool POS, int ND, bool K3>
__global__ __launch_bounds__(SMH_BLOCK_THREADS) void hash_kernel(smh_hash_ctx C, const uint32_t *__restrict__ bloom_g, uint64_t *count,
                                                                smh_pos_out po, smh_stats_arg SA)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(bloom_g);
        uint4 *dst = reinterpret_cast<uint4 *>(smh_lds);
        for (uint32_t i = threadIdx.x; i < C.P.bloom_bytes / 16u; i += SMH_BLOCK_THREADS) dst[i] = src[i];
    }
    const smh_chunk_sched S = smh_sched_init(smh_lds, C.P.bloom_bytes + SMH_HASH_QUEUES);
    smh_stats_stash(S.ctr_off, SA);
    __syncthreads();
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t *queue = reinterpret_cast<uint32_t *>(smh_lds + C.P.bloom_bytes) + (threadIdx.x >> 6) * SMH_HASH_QCAP;
    uint32_t events = 0;
    const uint32_t cnt = smh_hash_thread<POS, ND, K3>(gthread, S, C, smh_lds, queue, &po, &events);
    if constexpr (!POS) smh_block_finish(cnt, count, smh_lds, S.ctr_off, C.n, events); /* positions mode: the cursor is the count */
}

template <bool POS, int ND, bool K3 = false>
static hipError_t launch(const smh_hash_launch &L, hipStream_t stream)
{
    auto kern = hash_kernel<POS, ND, K3>;
    const uint32_t lds = L.C.P.bloom_bytes + SMH_HASH_QUEUES + 16u + SMH_SCHED_LDS;
    static smh_attr_cache cache;
    int per_cu = 0;
    const hipError_t err = cache.get(kern, lds, SMH_BLOCK_THREADS, &per_cu);
    if (err != hipSuccess) return err;
    per_cu = 1; /* one workgroup of 16 waves per CU streams best (wm_kernels.inc launch_pair) */
    const uint64_t chunk = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (L.C.n + chunk - 1) / chunk;
    uint64_t blocks = (uint64_t)L.n_cus * (uint64_t)per_cu;
    const uint64_t want = (n_chunks + (SMH_BLOCK_THREADS / 64) - 1) / (SMH_BLOCK_THREADS / 64);
    if (blocks > want) blocks = want;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SMH_BLOCK_THREADS), lds, stream, L.C, L.d_bloom, L.d_count, L.po, L.stats);
    return hipGetLastError();
}

/* ND = the window's dwords, (m + 3) / 4 = 1 .. 8: a template value (hash_lane.h smh_hash_verify) */
template <bool POS>
static hipError_t launch_nd(const smh_hash_launch &L, hipStream_t stream)
{
    if constexpr (!POS) {
        if (L.C.P.bloom_k >= 3u) /* the counting kernels test the filter's third bit (positions: two, stage 2 decides either way) */
            switch ((L.C.P.m + 3) / 4) {
            case 1: return launch<POS, 1, true>(L, stream);
            case 2: return launch<POS, 2, true>(L, stream);
            case 3: return launch<POS, 3, true>(L, stream);
            case 4: return launch<POS, 4, true>(L, stream);
            case 5: return launch<POS, 5, true>(L, stream);
            case 6: return launch<POS, 6, true>(L, stream);
            case 7: return launch<POS, 7, true>(L, stream);
            default: return launch<POS, 8, true>(L, stream);
            }
    }
    switch ((L.C.P.m + 3) / 4) {
    case 1: return launch<POS, 1>(L, stream);
    case 2: return launch<POS, 2>(L, stream);
    case 3: return launch<POS, 3>(L, stream);
    case 4: return launch<POS, 4>(L, stream);
    case 5: return launch<POS, 5>(L, stream);
    case 6: return launch<POS, 6>(L, stream);
    case 7: return launch<POS, 7>(L, stream);
    default: return launch<POS, 8>(L, stream);
    }
}
hipError_t smh_launch_hash(const smh_hash_launch &L, hipStream_t stream) { return launch_nd<false>(L, stream); }
hipError_t smh_launch_hash_positions(const smh_hash_launch &L, hipStream_t stream) { return launch_nd<true>(L, stream); }
