/*
 * csrc/wm_kernels.hip -- Wu-Manber scan kernels for gfx950 (MI355X).
 *
 * wm_block_kernel  tuned path: the device SHIFT table (block filter bit set) is
 *                  staged in LDS once per workgroup; every END column of a
 *                  lane's 64-byte segment is tested against it with a rolling
 *                  block code; survivors go through the HBM HASH/PREFIX verify
 *                  table.  Replaces wm_kernel3..5 (cuda/cuda_wm.cu:60-650).
 * wm_table_kernel  the reference tables as given: SHIFT staged in LDS
 *                  (16-bit), per-lane skip loop, CSR bucket scan, byte compare.
 *                  Replaces wm_kernel1/2 (cuda/cuda_wm.cu:786-1058).
 *
 * Roofline: HBM read, 1 byte per text symbol.  No MFMA.
 */
#include <hip/hip_runtime.h>
#include "smh_launch.h"
#include "wm_lane.h"

__device__ __forceinline__ void smh_wave_add_wm(uint32_t cnt, uint64_t *count)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    if ((threadIdx.x & 63u) == 0 && cnt) atomicAdd((unsigned long long *)count, (unsigned long long)cnt);
}

/* one atomic per workgroup instead of one per wave: see smh_block_add in ac_kernels.hip */
__device__ __forceinline__ void smh_block_add_wm(uint32_t cnt, uint64_t *count, unsigned char *lds)
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

template <bool HASHED, bool EXACT, int HC, int FK = 0, bool POS = false>
__global__ __launch_bounds__(SMH_BLOCK_THREADS) void wm_block_kernel(
    const uint8_t *__restrict__ text, uint64_t n, const uint32_t *__restrict__ filter_g, uint32_t lds_bytes,
    smh_wm_params P, int block_symbols, uint64_t *count, smh_pos_out po)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(filter_g);
        uint4 *dst = reinterpret_cast<uint4 *>(smh_lds);
        for (uint32_t i = threadIdx.x; i < lds_bytes / 16u; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const uint32_t *filter = reinterpret_cast<const uint32_t *>(smh_lds);
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    /* survivor queues: 1 KiB per wave right behind the filter */
    uint64_t *queue = EXACT ? nullptr
                            : reinterpret_cast<uint64_t *>(smh_lds + lds_bytes) + (threadIdx.x >> 6) * SMH_WM_QCAP;
    const uint32_t cnt = smh_wm_thread<HASHED, EXACT, HC, FK, POS>(gthread, nthreads, text, n, filter, P, block_symbols, queue, &po);
    if constexpr (!POS) smh_block_add_wm(cnt, count, smh_lds); /* positions mode: the cursor is the count */
}

/* alphabet 4, m <= 8: pair filter (two end columns per LDS lookup), 64 KiB of LDS */
template <bool POS>
__global__ __launch_bounds__(SMH_BLOCK_THREADS) void wm_pair_kernel(const uint8_t *__restrict__ text, uint64_t n, int m,
                                                                   const uint32_t *__restrict__ pair_g,
                                                                   const uint32_t *__restrict__ filter_g,
                                                                   uint64_t *count, smh_pos_out po)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(pair_g);
        uint4 *dst = reinterpret_cast<uint4 *>(smh_lds);
        /* 65536 bytes = 4096 x 16 B = 4 per thread, all four loads in flight */
        const uint4 t0 = src[threadIdx.x], t1 = src[threadIdx.x + 1024], t2 = src[threadIdx.x + 2048],
                    t3 = src[threadIdx.x + 3072];
        dst[threadIdx.x] = t0;
        dst[threadIdx.x + 1024] = t1;
        dst[threadIdx.x + 2048] = t2;
        dst[threadIdx.x + 3072] = t3;
    }
    __syncthreads();
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t cnt = smh_wm_pair_thread<true, POS>(gthread, nthreads, text, n, m, smh_lds, filter_g, &po);
    if constexpr (!POS) smh_block_add_wm(cnt, count, smh_lds);
}

__global__ __launch_bounds__(256) void wm_table_kernel(const uint8_t *__restrict__ text, uint64_t n, int m,
                                                      const uint16_t *__restrict__ shift_g, uint32_t shiftsize,
                                                      const uint32_t *__restrict__ bucket_off,
                                                      const int32_t *__restrict__ bucket,
                                                      const uint8_t *__restrict__ pat_orig, uint64_t *count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    uint16_t *shift = reinterpret_cast<uint16_t *>(smh_lds);
    for (uint32_t i = threadIdx.x; i < shiftsize; i += blockDim.x) shift[i] = shift_g[i];
    __syncthreads();
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t cnt = smh_wm_table_thread<uint16_t>(gthread, nthreads, text, n, shift, shiftsize, bucket_off,
                                                       bucket, pat_orig, m, 2);
    smh_wave_add_wm(cnt, count);
}

__global__ __launch_bounds__(256) void wm_positions_kernel(const uint8_t *__restrict__ text, uint64_t n, int m,
                                                          const uint16_t *__restrict__ shift_g, uint32_t shiftsize,
                                                          const uint32_t *__restrict__ bucket_off,
                                                          const int32_t *__restrict__ bucket,
                                                          const uint8_t *__restrict__ pat_orig, uint64_t *positions,
                                                          uint64_t capacity, uint64_t *cursor)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smh_lds[];
    uint16_t *shift = reinterpret_cast<uint16_t *>(smh_lds);
    for (uint32_t i = threadIdx.x; i < shiftsize; i += blockDim.x) shift[i] = shift_g[i];
    __syncthreads();
    const uint64_t gthread = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    smh_wm_positions_thread<uint16_t>(gthread, nthreads, text, n, shift, shiftsize, bucket_off, bucket, pat_orig, m, 2,
                                      positions, capacity, cursor);
}

hipError_t smh_launch_wm_positions(const smh_wm_table_launch &L, uint64_t *d_positions, uint64_t capacity,
                                   uint64_t *d_cursor, hipStream_t stream)
{
    const uint64_t per_block = 256ull * SMH_SEG;
    uint64_t blocks = (L.n + per_block - 1) / per_block;
    const uint64_t cap = (uint64_t)L.n_cus * 8u;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const uint32_t lds = (L.shiftsize * 2u + 15u) & ~15u;
    hipLaunchKernelGGL(wm_positions_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, L.d_text, L.n, L.m,
                       L.d_shift, L.shiftsize, L.d_bucket_off, L.d_bucket, L.d_pat_orig, d_positions, capacity,
                       d_cursor);
    return hipGetLastError();
}

/* ------------------------------------------------------------------ launch */
uint32_t smh_wm_max_blocks(int n_cus) { return (uint32_t)n_cus * 2u; }

template <bool HASHED, bool EXACT, int HC, int FK = 0, bool POS = false>
static hipError_t launch_one(const smh_wm_launch &L, hipStream_t stream)
{
    auto kern = wm_block_kernel<HASHED, EXACT, HC, FK, POS>;
    uint32_t lds_bytes = (uint32_t)(((uint64_t)1 << L.filter_log2) / 8u);
    if (lds_bytes < 16u) lds_bytes = 16u;
    const uint32_t lds_total = lds_bytes + (EXACT ? 0u : (SMH_BLOCK_THREADS / 64) * SMH_WM_QCAP * 8u);
    static uint32_t cached_lds = 0xFFFFFFFFu;
    static int cached_per_cu = 0;
    if (cached_lds != lds_bytes) { /* once per (kernel, LDS size): see ac_kernels.hip */
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_total);
        if (err != hipSuccess) return err;
        int q = 0;
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, SMH_BLOCK_THREADS, lds_total);
        if (err != hipSuccess) return err;
        cached_per_cu = q;
        cached_lds = lds_bytes;
    }
    int per_cu = cached_per_cu;
    if (per_cu < 1) per_cu = 1;
    const uint64_t chunk = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (L.n + chunk - 1) / chunk;
    uint64_t blocks = (uint64_t)L.n_cus * (uint64_t)per_cu;
    const uint64_t want = (n_chunks + (SMH_BLOCK_THREADS / 64) - 1) / (SMH_BLOCK_THREADS / 64);
    if (blocks > want) blocks = want;
    if (blocks < 1) blocks = 1;
    smh_wm_params P;
    P.m = L.m;
    P.bits = L.bits;
    const int wbits = L.block_symbols * L.bits;
    P.code_mask = wbits >= 32 ? 0xFFFFFFFFu : ((1u << wbits) - 1u);
    P.filter_log2 = L.filter_log2;
    P.filter_k = L.filter_k;
    P.filter_le4 = L.filter_le4;
    P.verify_log2 = L.verify_log2;
    P.verify = L.d_verify;
    P.pat_sorted = L.d_pat_sorted;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(SMH_BLOCK_THREADS), lds_total, stream, L.d_text, L.n,
                       L.d_filter, lds_bytes, P, L.block_symbols, L.d_count, L.po);
    return hipGetLastError();
}

template <bool HASHED, bool EXACT, bool POS>
static hipError_t launch_halo(const smh_wm_launch &L, hipStream_t stream)
{
    const int halo = L.m - 1;
    if constexpr (HASHED && !EXACT) {
        /* byte symbols, 4-byte block: the specialised scan (compile-time bits per key) */
        if (L.filter_le4 && halo <= 32) {
            if (halo <= 16) {
                if (L.filter_k == 2) return launch_one<true, false, 1, 2, POS>(L, stream);
                if (L.filter_k == 3) return launch_one<true, false, 1, 3, POS>(L, stream);
                return launch_one<true, false, 1, 4, POS>(L, stream);
            }
            if (L.filter_k == 2) return launch_one<true, false, 2, 2, POS>(L, stream);
            if (L.filter_k == 3) return launch_one<true, false, 2, 3, POS>(L, stream);
            return launch_one<true, false, 2, 4, POS>(L, stream);
        }
    }
    if (halo <= 16) return launch_one<HASHED, EXACT, 1, 0, POS>(L, stream);
    if (halo <= 32) return launch_one<HASHED, EXACT, 2, 0, POS>(L, stream);
    if (halo <= 64) return launch_one<HASHED, EXACT, 4, 0, POS>(L, stream);
    return launch_one<HASHED, EXACT, 0, 0, POS>(L, stream);
}

template <bool POS>
static hipError_t launch_pair(const smh_wm_launch &L, hipStream_t stream)
{
    auto wm_pair_kernel = ::wm_pair_kernel<POS>;
    const uint32_t lds_bytes = 65536u;
    static int cached_per_cu = 0;
    if (!cached_per_cu) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(wm_pair_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (err != hipSuccess) return err;
        int q = 0;
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, wm_pair_kernel, SMH_BLOCK_THREADS, lds_bytes);
        if (err != hipSuccess) return err;
        cached_per_cu = q < 1 ? 1 : q;
    }
    const uint64_t chunk = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (L.n + chunk - 1) / chunk;
    uint64_t blocks = (uint64_t)L.n_cus * (uint64_t)cached_per_cu;
    const uint64_t want = (n_chunks + (SMH_BLOCK_THREADS / 64) - 1) / (SMH_BLOCK_THREADS / 64);
    if (blocks > want) blocks = want;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wm_pair_kernel, dim3((unsigned)blocks), dim3(SMH_BLOCK_THREADS), lds_bytes, stream, L.d_text,
                       L.n, L.m, L.d_pair, L.d_filter, L.d_count, L.po);
    return hipGetLastError();
}

template <bool POS>
static hipError_t launch_block(const smh_wm_launch &L, hipStream_t stream)
{
    if (L.d_pair) return launch_pair<POS>(L, stream);
    if (L.filter_hashed) return launch_halo<true, false, POS>(L, stream);
    if (L.filter_exact) return launch_halo<false, true, POS>(L, stream);
    return launch_halo<false, false, POS>(L, stream);
}

hipError_t smh_launch_wm_block(const smh_wm_launch &L, hipStream_t stream) { return launch_block<false>(L, stream); }
/* positions mode of the same kernels (L.po set): END columns of all matches appended per wave */
hipError_t smh_launch_wm_block_positions(const smh_wm_launch &L, hipStream_t stream) { return launch_block<true>(L, stream); }

hipError_t smh_launch_wm_table(const smh_wm_table_launch &L, hipStream_t stream)
{
    const uint64_t per_block = 256ull * SMH_WM_TABLE_SPAN;
    uint64_t blocks = (L.n + per_block - 1) / per_block;
    const uint64_t cap = (uint64_t)L.n_cus * 8u;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const uint32_t lds = (L.shiftsize * 2u + 15u) & ~15u;
    hipLaunchKernelGGL(wm_table_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, L.d_text, L.n, L.m,
                       L.d_shift, L.shiftsize, L.d_bucket_off, L.d_bucket, L.d_pat_orig, L.d_count);
    return hipGetLastError();
}

/* ------------------------------------------------------------------ corpus */
__global__ void corpus_text_kernel(uint8_t *out, uint64_t n, uint64_t offset, uint64_t seed, uint32_t alphabet)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * 16u;
    for (uint64_t base = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16u; base < n; base += stride) {
        uint32_t v[4] = {0, 0, 0, 0};
        const uint64_t lim = n - base < 16u ? n - base : 16u;
        for (uint32_t k = 0; k < (uint32_t)lim; ++k) {
            uint64_t z = seed + (offset + base + k + 1) * 0x9E3779B97F4A7C15ULL;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            const uint32_t sym = (alphabet & (alphabet - 1)) == 0 ? (uint32_t)z & (alphabet - 1)
                                                                 : (uint32_t)(z % (uint64_t)alphabet);
            v[k >> 2] |= sym << (8 * (k & 3));
        }
        if (lim == 16u) {
            *reinterpret_cast<uint4 *>(out + base) = make_uint4(v[0], v[1], v[2], v[3]);
        } else {
            for (uint32_t k = 0; k < (uint32_t)lim; ++k) out[base + k] = (uint8_t)(v[k >> 2] >> (8 * (k & 3)));
        }
    }
}

hipError_t smh_launch_corpus_text(uint8_t *d_out, uint64_t n, uint64_t offset, uint64_t seed, int alphabet,
                                  hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n / 16u + 255u) / 256u;
    if (blocks > 8192u) blocks = 8192u;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(corpus_text_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, d_out, n, offset, seed,
                       (uint32_t)alphabet);
    return hipGetLastError();
}
