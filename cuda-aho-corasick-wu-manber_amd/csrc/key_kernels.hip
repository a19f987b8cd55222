/*
 * csrc/key_kernels.hip -- the key engine's kernels for gfx950 (key_hash.h, key_host.c, key_lane.h).
 *
 * The set's cuckoo image (two tables of 32- or 64-bit keys, up to 156 KiB) is staged in LDS once per 1024-thread
 * workgroup; the text streams through registers in 4 KiB wave-chunks taken from the workgroup's LDS counter
 * (lane_common.h); every END column costs two independent LDS reads and two compares.  No verify stage, no queue, no
 * dependence on the text: the one-pass exact engine for sets whose automaton does not fit LDS (replaces the walk of
 * ac/ac.c:207-219 / cuda/cuda_ac.cu:86-95 for them).  Roofline: HBM read, 1 byte per text symbol; bound in practice by
 * the VALU issue rate of the two hashes and by the LDS lookup rate.
 */
#include <hip/hip_runtime.h>
#include "smh_launch.h"
#include "key_lane.h"
#include "smh_stats.h"

template <int KC, int HP, bool POS, bool FULL>
__global__ __launch_bounds__(SMH_BLOCK_THREADS) void key_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                               const uint32_t *__restrict__ image_g, smh_key_params K,
                                                               uint64_t *count, smh_pos_out po, smh_stats_arg SA)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(image_g);
        uint4 *dst = reinterpret_cast<uint4 *>(smh_lds);
        for (uint32_t i = threadIdx.x; i < K.bytes / 16u; i += SMH_BLOCK_THREADS) dst[i] = src[i];
    }
    const smh_chunk_sched S = smh_sched_init(smh_lds, K.bytes < SMH_LDS_MIN ? SMH_LDS_MIN : K.bytes);
    smh_stats_stash(S.ctr_off, SA);
    __syncthreads();
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t cnt = smh_key_thread<KC, HP, POS, FULL>(gthread, S, text, n, smh_lds, K, &po);
    if constexpr (!POS) smh_block_finish(cnt, count, smh_lds, S.ctr_off, n, 0u); /* positions mode: the cursor is the count */
}

template <int KC, int HP, bool POS, bool FULL>
static hipError_t launch(const smh_key_launch &L, hipStream_t stream)
{
    auto kern = key_kernel<KC, HP, POS, FULL>;
    const uint32_t lds = (L.K.bytes < SMH_LDS_MIN ? SMH_LDS_MIN : L.K.bytes) + 16u + SMH_SCHED_LDS;
    static smh_attr_cache cache;
    int per_cu = 0;
    const hipError_t err = cache.get(kern, lds, SMH_BLOCK_THREADS, &per_cu);
    if (err != hipSuccess) return err;
    /* one workgroup of 16 waves per CU streams best (wm_kernels.inc launch_pair) */
    per_cu = per_cu < 1 ? 1 : (per_cu > L.wg_per_cu && L.wg_per_cu > 0 ? L.wg_per_cu : (L.wg_per_cu > 0 ? per_cu : 1));
    const uint64_t chunk = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (L.n + chunk - 1) / chunk;
    uint64_t blocks = (uint64_t)L.n_cus * (uint64_t)per_cu;
    const uint64_t want = (n_chunks + (SMH_BLOCK_THREADS / 64) - 1) / (SMH_BLOCK_THREADS / 64);
    if (blocks > want) blocks = want;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SMH_BLOCK_THREADS), lds, stream, L.d_text, L.n, L.d_image, L.K,
                       L.d_count, L.po, L.stats);
    return hipGetLastError();
}

/* round 6: the bucket image (key_hash.h): one 8-byte LDS read per column, an overflow table behind a wave-uniform branch */
template <int R, int HP, bool POS>
__global__ __launch_bounds__(SMH_BLOCK_THREADS) void keyb_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                                const uint32_t *__restrict__ image_g, smh_key_params K,
                                                                uint64_t *count, smh_pos_out po, smh_stats_arg SA)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(image_g);
        uint4 *dst = reinterpret_cast<uint4 *>(smh_lds);
        for (uint32_t i = threadIdx.x; i < K.bytes / 16u; i += SMH_BLOCK_THREADS) dst[i] = src[i];
    }
    /* LDS: the image | the waves' overflow queues (key_lane.h) | the chunk counter */
    const uint32_t image_bytes = K.bytes < SMH_LDS_MIN ? SMH_LDS_MIN : K.bytes;
    const smh_chunk_sched S = smh_sched_init(smh_lds, image_bytes + SMH_KEYB_QBYTES(R));
    smh_stats_stash(S.ctr_off, SA);
    __syncthreads();
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t queue_off = image_bytes + (threadIdx.x >> 6) * (SMH_KEYB_QBYTES(R) / 16u);
    const uint32_t cnt = smh_keyb_thread<R, HP, POS>(gthread, S, text, n, smh_lds, K, &po, queue_off);
    if constexpr (!POS) smh_block_finish(cnt, count, smh_lds, S.ctr_off, n, 0u);
}

template <int R, int HP, bool POS>
static hipError_t launch_bucket(const smh_key_launch &L, hipStream_t stream)
{
    auto kern = keyb_kernel<R, HP, POS>;
    const uint32_t lds = (L.K.bytes < SMH_LDS_MIN ? SMH_LDS_MIN : L.K.bytes) + SMH_KEYB_QBYTES(R) + 16u + SMH_SCHED_LDS;
    static smh_attr_cache cache;
    int per_cu = 0;
    const hipError_t err = cache.get(kern, lds, SMH_BLOCK_THREADS, &per_cu);
    if (err != hipSuccess) return err;
    per_cu = per_cu < 1 ? 1 : (per_cu > L.wg_per_cu && L.wg_per_cu > 0 ? L.wg_per_cu : (L.wg_per_cu > 0 ? per_cu : 1));
    const uint64_t chunk = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (L.n + chunk - 1) / chunk;
    uint64_t blocks = (uint64_t)L.n_cus * (uint64_t)per_cu;
    const uint64_t want = (n_chunks + (SMH_BLOCK_THREADS / 64) - 1) / (SMH_BLOCK_THREADS / 64);
    if (blocks > want) blocks = want;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SMH_BLOCK_THREADS), lds, stream, L.d_text, L.n, L.d_image, L.K,
                       L.d_count, L.po, L.stats);
    return hipGetLastError();
}

template <bool POS>
static hipError_t launch_any(const smh_key_launch &L, hipStream_t stream)
{
    if (L.K.layout == 1) {
        const bool hp2b = L.K.m - 1 > 16;
        if (L.K.bk_old == 15) return launch_bucket<15, 2, POS>(L, stream); /* alphabet 4, 17..22 symbols: the halo is always 32 bytes */
        if (L.K.bk_old == 6) return launch_bucket<6, 1, POS>(L, stream);   /* 20 letters, 7 or 8 symbols */
        return hp2b ? launch_bucket<0, 2, POS>(L, stream) : launch_bucket<0, 1, POS>(L, stream);
    }
    const bool hp2 = L.K.m - 1 > 16;
    const int kb = L.K.m * L.K.bits;
    /* the key fills its slot (alphabet 4: m = 16 / m = 32, the BASELINE lengths): no mask per column; counting kernels only */
    if (!POS && kb == 64) return hp2 ? launch<1, 2, POS, true>(L, stream) : launch<1, 1, POS, true>(L, stream);
    if (!POS && kb == 32) return hp2 ? launch<0, 2, POS, true>(L, stream) : launch<0, 1, POS, true>(L, stream);
    if (L.K.wide == 2) return hp2 ? launch<2, 2, POS, false>(L, stream) : launch<2, 1, POS, false>(L, stream); /* quotient keys */
    if (L.K.wide == 1) return hp2 ? launch<1, 2, POS, false>(L, stream) : launch<1, 1, POS, false>(L, stream);
    return hp2 ? launch<0, 2, POS, false>(L, stream) : launch<0, 1, POS, false>(L, stream);
}

hipError_t smh_launch_keys(const smh_key_launch &L, hipStream_t stream) { return launch_any<false>(L, stream); }
hipError_t smh_launch_keys_positions(const smh_key_launch &L, hipStream_t stream) { return launch_any<true>(L, stream); }
