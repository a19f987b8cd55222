/*
 * csrc/smh_runtime.hip -- the C-ABI shim between the C host code and the HIP kernels.
 *
 *   - device-side copies of the compiled tables (uploaded lazily, once per handle)
 *   - smh_ac_scan / smh_wm_scan: asynchronous launches on the caller's stream
 *   - the blocking *_count_host helpers (upload text, scan, download the count)
 *   - the legacy entry points with the reference's shapes: search_ac, search_wu,
 *     search_wu2 (smatcher.h:90,105-106) and cuda_ac1..5 / cuda_wm1..5
 *     (cuda/cuda_ac.cu:594-1072, cuda/cuda_wm.cu:183-1180)
 *
 * There is no CPU search path here: when no HIP device is usable the extended
 * API returns SMH_ENODEV and the legacy names print the reason and exit(1).
 */
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <math.h>
#include <mutex>
#include <thread>
#include <vector>
#include "smh_internal.h"
#include "smh_launch.h"
#include "smh_stats.h"

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            smh_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SMH_ENODEV;                                                            \
        }                                                                                 \
    } while (0)

struct smh_ac_dev {
    int device;
    smh_ac_dev *next; /* the handle keeps one table set per device it has scanned on */
    void *d_table;
    void *d_scan;
    void *d_trunc1;
    uint64_t *d_queue;
    uint32_t *d_depth_first;
    smh_ac_cold_ctx *d_cold; /* what the kernels' rare paths read (ac_lane.h), pointers into this set and the hv handle's */
    uint32_t *d_dense_pair;   /* dense plan (smh_internal.h): 64 KiB pair bit set, and the plain 4^m-bit set */
    uint32_t *d_dense_filter;
    int32_t *d_transition;
    uint32_t *d_supply;
    uint32_t *d_final;
};

struct smh_wm_dev {
    int device;
    smh_wm_dev *next;
    uint32_t *d_filter;
    uint32_t *d_pair;
    uint32_t *d_gram;
    uint64_t *d_queue;
    uint32_t *d_verify;
    uint32_t *d_verify_ck;
    uint8_t *d_pat_sorted;
    uint16_t *d_shift;
    uint32_t *d_bucket_off;
    int32_t *d_bucket;
    uint8_t *d_pat_orig;
    smh_wm_class *d_classes; /* SMH_WM_MAX_CLASSES entries, when the handle is the suffix filter of a mixed-length set */
    smh_wm_class h_classes[SMH_WM_MAX_CLASSES]; /* what d_classes holds (wm_multi_launch uploads only what has changed) */
    int n_classes_up;
};

/* Device-side state is kept PER DEVICE: a handle owns one table set for every device it has been
 * scanned on (a single process that drives all GPUs of a node -- smh_multi.hip, `smatcher -ranks R` --
 * switches devices between launches and must not re-upload).  The lists and the CU-count cache are
 * guarded by one mutex; launches themselves run outside it. */
static std::mutex g_dev_mu;
static int g_n_cus[SMH_MAX_DEVICES];

static int current_cus(int *n_cus)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_dev_mu);
    if (dev < 0 || dev >= SMH_MAX_DEVICES || g_n_cus[dev] == 0) {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        const int v = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        if (dev < 0 || dev >= SMH_MAX_DEVICES) { *n_cus = v; return SMH_OK; }
        g_n_cus[dev] = v;
    }
    *n_cus = g_n_cus[dev];
    return SMH_OK;
}

/* One process may drive several LOGICAL devices that share a card (smh_multi.hip with SMH_MULTI_SHARE_DEVICE: the
 * one-card rehearsal of the N-device flow): the calling thread names its logical slot and every per-device list of
 * this file is keyed by (device, slot), so each logical shard has a table set, a candidate-queue workspace and an
 * adaptive state of its own, exactly as on N cards.  Slot 0 = the plain case. */
static thread_local int g_dev_slot = 0;
extern "C" void smh_dev_set_slot(int slot) { g_dev_slot = slot < 0 ? 0 : slot; }
static int current_dev_key(int *key)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    *key = dev | (g_dev_slot << 8);
    return SMH_OK;
}

/* find the table set of the current device in a handle's list, or build one with `build` and publish it
 * only when every upload succeeded (a half-built set is freed, so the next call retries cleanly).  The build
 * -- synchronous hipMalloc + hipMemcpy of the tables -- runs OUTSIDE the mutex: a process that drives every
 * GPU of a node prepares its devices from one thread each (smh_multi_*_prepare) and those uploads must run
 * side by side; the list is only searched and extended under the lock.  A second thread that wants the set of the
 * SAME list and device while it is being built waits for it (round 4: it used to build a second copy and free the
 * loser's -- transiently twice the device memory of a multi-GB table set). */
#ifdef SMH_TESTING
static std::atomic<int> g_builds_now{0}, g_builds_peak{0};
extern "C" int smh_dev_build_peak(int reset) /* test hook: most table-set builds ever in flight together */
{
    const int v = g_builds_peak.load();
    if (reset) g_builds_peak.store(0);
    return v;
}
#endif
struct smh_building { const void *head; int key; };
static std::vector<smh_building> g_building; /* guarded by g_dev_mu */
static std::condition_variable g_building_cv;
template <typename D, typename Build>
static int ensure_device_set(D **head, void (*free_one)(D *), Build build, D **out)
{
    int dev = 0;
    int rc0 = current_dev_key(&dev);
    if (rc0 != SMH_OK) return rc0;
    {
        std::unique_lock<std::mutex> lock(g_dev_mu);
        for (;;) {
            for (D *d = *head; d; d = d->next)
                if (d->device == dev) { *out = d; return SMH_OK; }
            bool busy = false;
            for (const smh_building &b : g_building) busy = busy || (b.head == (const void *)head && b.key == dev);
            if (!busy) break;
            g_building_cv.wait(lock); /* another thread is building exactly this set: take its result (or retry after its failure) */
        }
        g_building.push_back(smh_building{(const void *)head, dev});
    }
    D *d = new D();
    memset(d, 0, sizeof *d);
    d->device = dev;
#ifdef SMH_TESTING
    const int now = ++g_builds_now;
    for (int peak = g_builds_peak.load(); now > peak && !g_builds_peak.compare_exchange_weak(peak, now);) {}
    if (const char *e = getenv("SMH_TEST_BUILD_DELAY_MS")) /* test hook: makes "two builds overlap" deterministic */
        std::this_thread::sleep_for(std::chrono::milliseconds(atoi(e)));
#endif
    const int rc = build(d);
#ifdef SMH_TESTING
    --g_builds_now;
#endif
    {
        std::lock_guard<std::mutex> lock(g_dev_mu);
        for (size_t i = 0; i < g_building.size(); ++i)
            if (g_building[i].head == (const void *)head && g_building[i].key == dev) { g_building.erase(g_building.begin() + (long)i); break; }
        if (rc == SMH_OK) {
            d->next = *head;
            *head = d;
        }
    }
    g_building_cv.notify_all();
    if (rc != SMH_OK) { free_one(d); return rc; }
    *out = d;
    return SMH_OK;
}

/* development aid (tools/wavetrace.py, not in the public headers): when set, the tuned AC kernel stores
 * three 100 MHz timestamps per wave (start, table staged, done) into this device buffer, which must
 * hold 3 * 16 * smh_ac_max_blocks(CUs) entries */
static uint64_t *g_wave_trace = NULL;
extern "C" void smh_dev_set_wave_trace(uint64_t *d_buf) { g_wave_trace = d_buf; }

/* ------------------------------------------------------------------ runtime wrappers */
extern "C" int smh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int smh_set_device(int device)
{
    HIP_TRY(hipSetDevice(device));
    return SMH_OK;
}

extern "C" int smh_device_name(char *buf, size_t cap)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    snprintf(buf, cap, "%s %s (%d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return SMH_OK;
}

extern "C" int smh_device_pci_bus_id(char *buf, size_t cap)
{
    if (!buf || cap < 16) { smh_set_error("smh_device_pci_bus_id: buffer of 16 bytes or more"); return SMH_EINVAL; }
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetPCIBusId(buf, (int)cap, dev));
    return SMH_OK;
}

extern "C" int smh_device_malloc(void **dptr, uint64_t bytes)
{
    if (!dptr) { smh_set_error("smh_device_malloc: NULL"); return SMH_EINVAL; }
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 16));
    return SMH_OK;
}

extern "C" int smh_device_free(void *dptr)
{
    if (dptr) HIP_TRY(hipFree(dptr));
    return SMH_OK;
}

extern "C" int smh_device_memset(void *dptr, int value, uint64_t bytes, void *stream)
{
    HIP_TRY(hipMemsetAsync(dptr, value, bytes, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_copy_to_device(void *dst, const void *src, uint64_t bytes, void *stream)
{
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_copy_to_host(void *dst, const void *src, uint64_t bytes, void *stream)
{
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_stream_synchronize(void *stream)
{
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return SMH_OK;
}

/* ---- measurement aid: what a pure streaming read of the same buffer reaches on this device.
 * SURVEY 8(d) asks for a read-only streaming kernel bandwidth from the same run beside the
 * roofline fraction; this is that kernel: 16-byte loads, four in flight per lane, grid-stride,
 * one XOR per load, one atomic per workgroup.  Not used by any scan path. */
__global__ __launch_bounds__(1024) void smh_stream_read_kernel(const uint4 *__restrict__ p, uint64_t n16,
                                                               unsigned long long *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 3u * stride < n16; i += 4u * stride) {
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2u * stride], d = p[i + 3u * stride];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) {
        const uint4 a = p[i];
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc ^= __shfl_down(acc, off, 64);
    __shared__ uint32_t part[16];
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t v = 0;
        for (unsigned w = 0; w < (blockDim.x >> 6); ++w) v ^= part[w];
        atomicXor(out, (unsigned long long)v);
    }
}

extern "C" int smh_stream_read_probe(const void *d_buf, uint64_t bytes, uint64_t *d_out, void *stream)
{
    if (!d_buf || !d_out || ((uintptr_t)d_buf & 15u) != 0) {
        smh_set_error("smh_stream_read_probe: bad arguments");
        return SMH_EINVAL;
    }
    int n_cus = 0, rc;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    hipLaunchKernelGGL(smh_stream_read_kernel, dim3((unsigned)n_cus * 2u), dim3(1024), 0, (hipStream_t)stream,
                       (const uint4 *)d_buf, bytes / 16u, (unsigned long long *)d_out);
    HIP_TRY(hipGetLastError());
    return SMH_OK;
}

/* More shapes of the same measurement.  The grid-stride probe above was beaten by real scan kernels (5.8 against 6.3
 * TB/s in round 2): "what a streaming read reaches here" depends on how much is in flight per CU.  tools/readsweep.hip
 * swept waves per CU x loads in flight x access shape on this device (gpurun_out/r03_b/readsweep.log): the best pure
 * reads keep about 32 KiB in flight per CU -- 8 waves x 4 KiB -- and reach 6.4-6.5 TB/s; 32 waves per CU lose 5-10 %,
 * non-temporal loads 40 %.  Every probe adds its result with ONE atomic per workgroup: the first 16-wave probes of round 3
 * used one per wave and read "5.5 TB/s" -- 25 us of same-address atomics at the end; with the reduction they read 6.43.
 * smh_stream_read_probe_variant(v):
 *   0  grid-stride, 16-byte loads, 4 in flight, 32 waves/CU (the round-2 probe)
 *   1  4 KiB wave-chunks (a lane's 64-byte segment as four 16-byte loads: the scan kernels' shape), dealt round-robin
 *      to 8 waves/CU (256 threads x 2 workgroups)
 *   2  grid-stride, 4 loads in flight, 8 waves/CU (512 threads x 1)
 *   3  4 KiB wave-chunks, two chunks in flight per wave, 4 waves/CU
 *   4  4 KiB wave-chunks taken from the workgroup's LDS counter at 16 waves/CU -- exactly how the scan kernels run
 * bench.py reports the best of them. */
typedef uint32_t smh_v4u __attribute__((ext_vector_type(4)));
/* XOR of a workgroup's values into *out with ONE atomic: one per wave is 4096 same-address atomics at ~12 ns each when the
 * waves finish together -- a 25 us tail on a 170 us probe, which is what made the 16-wave probes of this file look slow */
__device__ __forceinline__ void smh_probe_block_xor(uint32_t acc, unsigned long long *out, uint32_t *part /* 16 words of LDS */)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc ^= __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t v = 0;
        for (unsigned w = 0; w < (blockDim.x >> 6); ++w) v ^= part[w];
        if (v) atomicXor(out, (unsigned long long)v);
    }
}
template <int C>
__global__ __launch_bounds__(1024) void smh_stream_chunk_kernel(const uint8_t *__restrict__ text, uint64_t n_chunks, unsigned long long *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t acc = 0;
    uint64_t k = wave;
    for (; k + (uint64_t)(C - 1) * nw < n_chunks; k += (uint64_t)C * nw) {
        smh_v4u v[C][4];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[c][q] = *reinterpret_cast<const smh_v4u *>(text + (k + (uint64_t)c * nw) * 4096u + lane * 64u + 16u * q);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc ^= v[c][q].x ^ v[c][q].y ^ v[c][q].z ^ v[c][q].w;
    }
    for (; k < n_chunks; k += nw) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const smh_v4u t = *reinterpret_cast<const smh_v4u *>(text + k * 4096u + lane * 64u + 16u * q);
            acc ^= t.x ^ t.y ^ t.z ^ t.w;
        }
    }
    __shared__ uint32_t probe_part[16];
    smh_probe_block_xor(acc, out, probe_part);
}

__global__ __launch_bounds__(1024) void smh_stream_sched_kernel(const uint8_t *__restrict__ text, uint64_t n_chunks, unsigned long long *out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char probe_lds[];
    const smh_chunk_sched S = smh_sched_init(probe_lds, 0);
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t acc = 0;
    for (uint64_t k = S.take(n_chunks); k < n_chunks; k = S.take(n_chunks)) {
        const smh_v4u *p = reinterpret_cast<const smh_v4u *>(text + k * 4096u + (uint64_t)lane * 64u);
        smh_v4u v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = p[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
    }
    __syncthreads(); /* every wave is done with the chunk counter */
    smh_probe_block_xor(acc, out, reinterpret_cast<uint32_t *>(probe_lds + 64));
}

extern "C" int smh_stream_read_probe_variant(const void *d_buf, uint64_t bytes, uint64_t *d_out, void *stream, int variant)
{
    if (variant == 0) return smh_stream_read_probe(d_buf, bytes, d_out, stream);
    if (!d_buf || !d_out || ((uintptr_t)d_buf & 15u) != 0 || variant < 0 || variant > 4) {
        smh_set_error("smh_stream_read_probe_variant: bad arguments");
        return SMH_EINVAL;
    }
    int n_cus = 0, rc;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    const uint8_t *t = (const uint8_t *)d_buf;
    unsigned long long *o = (unsigned long long *)d_out;
    hipStream_t st = (hipStream_t)stream;
    if (variant == 1) {
        hipLaunchKernelGGL(smh_stream_chunk_kernel<1>, dim3((unsigned)n_cus * 2u), dim3(256), 0, st, t, bytes / 4096u, o);
    } else if (variant == 2) {
        hipLaunchKernelGGL(smh_stream_read_kernel, dim3((unsigned)n_cus), dim3(512), 0, st, (const uint4 *)d_buf, bytes / 16u, o);
    } else if (variant == 3) {
        hipLaunchKernelGGL(smh_stream_chunk_kernel<2>, dim3((unsigned)n_cus), dim3(256), 0, st, t, bytes / 4096u, o);
    } else {
        const uint32_t lds = 96u * 1024u; /* one workgroup per CU, as a scan kernel whose table fills LDS */
        static smh_attr_cache cache;
        int q = 0;
        HIP_TRY(cache.get(smh_stream_sched_kernel, lds, 1024, &q));
        hipLaunchKernelGGL(smh_stream_sched_kernel, dim3((unsigned)n_cus), dim3(1024), lds, st, t, bytes / 4096u, o);
    }
    HIP_TRY(hipGetLastError());
    return SMH_OK;
}

extern "C" int smh_corpus_text_device(unsigned char *d_out, uint64_t n, uint64_t offset, uint64_t seed,
                                      int alphabet, void *stream)
{
    if (!d_out || alphabet < 1 || alphabet > 256) { smh_set_error("smh_corpus_text_device: bad arguments"); return SMH_EINVAL; }
    if (((uintptr_t)d_out & 15u) != 0) { smh_set_error("smh_corpus_text_device: buffer must be 16-byte aligned"); return SMH_EINVAL; }
    HIP_TRY(smh_launch_corpus_text(d_out, n, offset, seed, alphabet, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_corpus_tabs_build(struct smh_corpus_tabs *T, uint64_t seed, int alphabet, int kind); /* corpus.c */
extern "C" int smh_corpus_text_device_kind(unsigned char *d_out, uint64_t n, uint64_t offset, uint64_t seed, int alphabet,
                                           int kind, void *stream)
{
    if (kind == SMH_CORPUS_UNIFORM) return smh_corpus_text_device(d_out, n, offset, seed, alphabet, stream);
    if (!d_out || ((uintptr_t)d_out & 15u) != 0) { smh_set_error("smh_corpus_text_device_kind: buffer must be 16-byte aligned"); return SMH_EINVAL; }
    smh_corpus_tabs T;
    const int rc = smh_corpus_tabs_build(&T, seed, alphabet, kind);
    if (rc != SMH_OK) return rc;
    HIP_TRY(smh_launch_corpus_text_kind(d_out, n, offset, seed, alphabet, kind, T, (hipStream_t)stream));
    return SMH_OK;
}

/* upload `bytes` of host data into a fresh device buffer padded to `pad_to` extra readable bytes */
static int upload(void **d, const void *h, size_t bytes, size_t pad)
{
    size_t total = ((bytes + pad + 15) / 16) * 16;
    if (total < 16) total = 16;
    HIP_TRY(hipMalloc(d, total));
    HIP_TRY(hipMemset(*d, 0, total));
    if (bytes) HIP_TRY(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return SMH_OK;
}

#include "smh_adapt.h" /* the adaptive engine: state, polling, policy (no HIP calls; also built for the CPU by tools/tsan_adapt.cpp) */

static void adapt_free_one(smh_adapt_dev *a)
{
    if (a->order_ev) (void)hipEventDestroy((hipEvent_t)a->order_ev);
    (void)hipFree(a->d_stats);
    (void)hipFree(a->d_scratch);
    if (a->h_rec) (void)hipHostFree(a->h_rec);
    delete a->mu;
    delete a;
}

extern "C" void smh_adapt_dev_free(struct smh_adapt_dev *a)
{
    while (a) {
        smh_adapt_dev *next = a->next;
        adapt_free_one(a);
        a = next;
    }
}

static int adapt_get(smh_adapt_dev **head, smh_adapt_dev **out)
{
    return ensure_device_set<smh_adapt_dev>(head, adapt_free_one, [&](smh_adapt_dev *a) -> int {
        a->engine = -1;
        a->mode_density = -1.0;
        a->slow = 1.0;
        a->mu = new std::mutex();
        const size_t rec_bytes = SMH_STATS_SLOTS * SMH_STATS_HOST_WORDS * sizeof(unsigned long long);
        HIP_TRY(hipHostMalloc((void **)&a->h_rec, rec_bytes, hipHostMallocDefault));
        memset(a->h_rec, 0, rec_bytes);
        smh_scan_stats init[SMH_STATS_SLOTS] = {};
        for (unsigned int i = 0; i < SMH_STATS_SLOTS; ++i) init[i].host = a->h_rec + i * SMH_STATS_HOST_WORDS; /* pinned host memory has one address on both sides */
        HIP_TRY(hipMalloc((void **)&a->d_stats, sizeof init));
        HIP_TRY(hipMemcpy(a->d_stats, init, sizeof init, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&a->d_scratch, 64));
        HIP_TRY(hipMemset(a->d_scratch, 0, 64));
        return SMH_OK;
    }, out);
}

/* the adaptive state of the current device if one exists (never creates) */
static smh_adapt_dev *adapt_find(smh_adapt_dev *const *head)
{
    int key = 0;
    if (current_dev_key(&key) != SMH_OK) return NULL;
    std::lock_guard<std::mutex> lock(g_dev_mu);
    for (smh_adapt_dev *a = *head; a; a = a->next)
        if (a->device == key) return a;
    return NULL;
}

/* Launches of ONE handle are ordered on the device, also across streams (round 5).  A handle owns per-device workspaces -- the
 * depth-cut automaton kernels' candidate queue, the report slots -- and its measured rates are only its own while its launches do
 * not run beside each other.  Same stream as the previous launch: in order anyway, nothing to do, nothing recorded (the
 * single-stream caller pays nothing).  The first time a second stream shows up nothing was recorded behind the earlier launches:
 * an event is recorded on the previous stream then (once per handle and device; the host does not block).  From then on every launch records an event behind itself and a launch
 * on another stream than the previous one waits for it on the device (the host never blocks).  Not inside a stream capture.
 * Called with A->mu held. */
static int adapt_order_before(smh_adapt_dev *A, void *stream)
{
    A->unordered = 0;
    if (!A->have_stream || A->last_stream == stream) return SMH_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        A->unordered = 1;
        return SMH_OK;
    }
    if (!A->multi_stream) {
        /* first change of stream: nothing has been recorded behind the earlier launches yet.  Round 6: an event recorded NOW on the
         * previous stream stands behind all of them; the new stream waits for it on the device.  (Round 5 did a hipDeviceSynchronize
         * here: it blocked the host with A->mu held and failed -- invalidating the capture -- when ANY stream of the process was
         * capturing in global mode.)  A previous stream that is capturing, or that the caller has destroyed since, cannot take the
         * record: this launch then goes unordered, as inside a capture, and its duration is not used. */
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        A->order_ev = (void *)ev;
        A->multi_stream = 1;
        hipStreamCaptureStatus prev = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)A->last_stream, &prev) != hipSuccess || prev != hipStreamCaptureStatusNone ||
            hipEventRecord(ev, (hipStream_t)A->last_stream) != hipSuccess) {
            (void)hipGetLastError();
            A->unordered = 1;
            return SMH_OK;
        }
    }
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)A->order_ev, 0));
    return SMH_OK;
}
static int adapt_order_after(smh_adapt_dev *A, void *stream)
{
    A->last_stream = stream;
    A->have_stream = 1;
    if (A->multi_stream && !A->unordered) HIP_TRY(hipEventRecord((hipEvent_t)A->order_ev, (hipStream_t)stream));
    return SMH_OK;
}

/* First look at a text (round 4).  The policy above needs a report to act on, so a handle's FIRST launch on a device runs the
 * compile's choice whatever the text -- on the planted corpus the headline's m = 32 set through the hybrid image: 33 ms for one
 * GiB that the plain stride-1 parts scan in 0.63.  The first tuned count launch of a handle on a device is synchronous anyway (its
 * table set goes up with blocking copies), so when it is a long one (1 GiB or more) it looks first: a 1 MiB launch to get the
 * engine's code and tables resident, then the text's first 256 MiB with a report, waited for.  Three times the estimate or more
 * means the text is not of the kind the estimates were made on: the other engines scan the same piece (into a scratch count,
 * twice each: the first launch of an engine runs cold) and the rest of the text goes to whichever measured best.  On ordinary
 * text that is two short launches and one stream synchronisation more, once per handle and device.  launch(engine, text, n,
 * count, SA) runs one engine; *done = the bytes whose END columns have been counted when the function returns. */
#define SMH_FIRST_LOOK_BYTES (256ull << 20) /* a 64 MiB piece is 11 us of streaming: launch overhead and table staging made uniform text look hostile and the engines' figures noise */
#define SMH_FIRST_LOOK_MIN_TEXT (1ull << 30)
template <typename LAUNCH>
static int adapt_first_look(smh_adapt_dev *A, const double est[SMH_ENGINES], int initial, int m, const unsigned char *d_text, uint64_t n,
                            uint64_t *d_count, void *stream, LAUNCH &&launch, uint64_t *done)
{
    *done = 0;
    /* (gated on "no report held yet", not on "no launch yet": a short warm-up scan -- smh_multi_*_prepare's 16 KiB one -- sets the
     * engine without telling anything about the text) */
    if (A->reports > 0 || n < SMH_FIRST_LOOK_MIN_TEXT || est[initial] <= 0) return SMH_OK;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return SMH_OK; }
    int others = 0;
    for (int o = 0; o < SMH_ENGINES; ++o) others += o != initial && est[o] > 0;
    if (!others) return SMH_OK;
    uint64_t *scratch = A->d_scratch; /* device memory nobody else reads */
    const uint64_t piece = SMH_FIRST_LOOK_BYTES + (uint64_t)(m - 1); /* END columns [m-1, piece) = the starts [0, 256 MiB) */
    auto probe = [&](int engine, uint64_t *count, uint64_t bytes, bool report) -> int {
        smh_stats_arg sa = {};
        if (report) sa = adapt_slot(A, bytes, engine, false);
        int rc = launch(engine, d_text, bytes, count, sa);
        if (rc != SMH_OK) return rc;
        HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
        adapt_poll(A);
        return SMH_OK;
    };
    int rc;
    A->engine = initial;
    ++A->launches;
    if ((rc = probe(initial, scratch, 1u << 20, false)) != SMH_OK) return rc; /* code and tables resident */
    if ((rc = probe(initial, d_count, piece, true)) != SMH_OK) return rc;
    *done = SMH_FIRST_LOOK_BYTES;
    const double c0 = adapt_ms(A, initial);
    if (A->n[initial] == 0 || c0 <= 3.0 * est[initial]) return SMH_OK;
    if (!engine_text_independent(initial) && c0 / est[initial] > A->slow) A->slow = c0 / est[initial];
    double floor_ms = 0.0; /* the best text-independent engine measured so far */
    for (int o = SMH_ENGINES - 1; o >= 0; --o) { /* the text-independent engines first: the others may be as slow as the first */
        if (o == initial || est[o] <= 0) continue;
        if (!engine_text_independent(o) && floor_ms > 0 && est[o] * A->slow > 2.0 * floor_ms) continue; /* no chance */
        if (engine_text_independent(o) && floor_ms > 0 && est[o] > 1.3 * floor_ms) continue; /* its estimate holds on any text */
        for (int k = 0; k < 2; ++k)
            if ((rc = probe(o, scratch, piece, true)) != SMH_OK) return rc;
        if (engine_text_independent(o) && adapt_ms(A, o) > 0 && (floor_ms == 0.0 || adapt_ms(A, o) < floor_ms)) floor_ms = adapt_ms(A, o);
    }
    int best = initial;
    for (int o = 0; o < SMH_ENGINES; ++o)
        if (A->n[o] > 0 && adapt_ms(A, o) * 1.03 < adapt_ms(A, best)) best = o;
    if (best != initial && adapt_ms(A, initial) < 2.0 * adapt_ms(A, best)) {
        /* a close call against a figure from the device's first busy milliseconds (clocks still ramping: DESIGN 6): the first
         * engine gets its second launch like the others before it is judged */
        if ((rc = probe(initial, scratch, piece, true)) != SMH_OK) return rc;
        best = initial;
        for (int o = 0; o < SMH_ENGINES; ++o)
            if (A->n[o] > 0 && adapt_ms(A, o) * 1.03 < adapt_ms(A, best)) best = o;
    }
    if (best != initial) {
        A->engine = best;
        A->ref_valid = 0;
        ++A->flips;
    }
    A->fresh = 0;
    return SMH_OK;
}

/* the key engine's launch (below, "key engine"): END columns of [d_text, d_text + n) counted into *d_count or appended to po */
static int keys_launch(struct smh_keys *k, const unsigned char *d_text, uint64_t n, uint64_t *d_count, const smh_pos_out *po,
                       void *stream, const smh_stats_arg &SA);
struct smh_keys_dev;
static int keys_ensure_device(struct smh_keys *k, smh_keys_dev **out);
/* the window-hash engine's launch (below) */
static int hash_launch(struct smh_hashes *k, const unsigned char *d_text, uint64_t n, uint64_t *d_count, const smh_pos_out *po,
                       void *stream, const smh_stats_arg &SA);
struct smh_hash_dev;
static int hash_ensure_device(struct smh_hashes *k, smh_hash_dev **out);
/* ------------------------------------------------------------------ AC */
static void ac_dev_free_one(smh_ac_dev *dev)
{
    (void)hipFree(dev->d_table);
    if (dev->d_trunc1 != dev->d_scan) (void)hipFree(dev->d_trunc1);
    (void)hipFree(dev->d_scan);
    (void)hipFree(dev->d_queue);
    (void)hipFree(dev->d_depth_first);
    (void)hipFree(dev->d_cold);
    (void)hipFree(dev->d_dense_pair);
    (void)hipFree(dev->d_dense_filter);
    (void)hipFree(dev->d_transition);
    (void)hipFree(dev->d_supply);
    (void)hipFree(dev->d_final);
    delete dev;
}

extern "C" void smh_ac_dev_free(struct smh_ac_dev *dev) /* the whole list */
{
    while (dev) {
        smh_ac_dev *next = dev->next;
        ac_dev_free_one(dev);
        dev = next;
    }
}

static int ac_ensure_device(struct smh_ac *ac, smh_ac_dev **out)
{
    int n_cus = 0, rc;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    return ensure_device_set<smh_ac_dev>(&ac->dev, ac_dev_free_one, [&](smh_ac_dev *d) -> int {
        int rc;
        /* a hybrid image may run with its full-row lookups left out of range: ask the device once, here, where blocking is
         * expected -- never inside a scan call (ac_kernels.inc) */
        if (ac->scan_full_rows) (void)smh_lds_oob_probe(n_cus);
        /* 256 entries of slack: a text byte >= alphabet may index just past the last row */
        if ((rc = upload(&d->d_table, ac->table, (size_t)ac->table_bytes, 256 * 4)) != SMH_OK) return rc;
        if ((rc = upload(&d->d_scan, ac->scan_table, (size_t)ac->scan_bytes, 0)) != SMH_OK) return rc;
        if (ac->trunc1_table == ac->scan_table) {
            d->d_trunc1 = d->d_scan;
        } else if ((rc = upload(&d->d_trunc1, ac->trunc1_table, (size_t)ac->trunc1_bytes, 256 * 4)) != SMH_OK) {
            return rc;
        }
        if (ac->scan_dense) {
            const size_t fbytes = ((size_t)1 << (2 * ac->m)) / 8;
            if ((rc = upload((void **)&d->d_dense_pair, ac->dense_pair, 65536, 0)) != SMH_OK) return rc;
            if ((rc = upload((void **)&d->d_dense_filter, ac->dense_filter, fbytes < 4 ? 4 : fbytes, 0)) != SMH_OK) return rc;
        }
        if (!ac->scan_exact) {
            const size_t qbytes = (size_t)smh_ac_max_blocks(n_cus) * (SMH_BLOCK_THREADS / 64) * SMH_AC_QCAP * 8;
            HIP_TRY(hipMalloc((void **)&d->d_queue, qbytes));
        }
        size_t dflen = (size_t)ac->m + 2;
        if (dflen < SMH_DEPTH_FIRST_MIN) dflen = SMH_DEPTH_FIRST_MIN;
        std::vector<uint32_t> df(dflen, ac->rows);
        for (int i = 0; i <= ac->max_depth + 1 && (size_t)i < dflen; ++i) df[i] = ac->depth_first[i];
        return upload((void **)&d->d_depth_first, df.data(), df.size() * 4, 0);
    }, out);
}

/* the reference-layout tables go up on the first SMH_VARIANT_TABLE scan only: for an alphabet-256
 * automaton they are the largest object the handle owns and the tuned kernel never reads them */
static int ac_ensure_reference_tables(struct smh_ac *ac, smh_ac_dev *d)
{
    std::lock_guard<std::mutex> lock(g_dev_mu);
    if (d->d_final) return SMH_OK; /* the last of the three to go up */
    if (!ac->g_transition) {
        smh_set_error("smh_ac_scan: this handle carries no reference-layout tables (it came from preproc_ac, "
                      "whose search_ac runs the tuned kernel only); build it with smh_ac_compile_tables or "
                      "smh_ac_compile_patterns for SMH_VARIANT_TABLE");
        return SMH_EUNSUP;
    }
    int rc;
    const size_t A = (size_t)ac->alphabet;
    int32_t *t = NULL;
    uint32_t *su = NULL, *fi = NULL;
    if ((rc = upload((void **)&t, ac->g_transition, (size_t)ac->states * A * 4, 0)) != SMH_OK ||
        (rc = upload((void **)&su, ac->g_supply, (size_t)ac->states * 4, 0)) != SMH_OK ||
        (rc = upload((void **)&fi, ac->g_final, (size_t)ac->states * 4, 0)) != SMH_OK) {
        (void)hipFree(t); (void)hipFree(su); (void)hipFree(fi);
        return rc;
    }
    d->d_transition = t; d->d_supply = su; d->d_final = fi;
    return SMH_OK;
}

/* everything a scan of this variant needs on the current device, without launching: the legacy
 * wrappers call it BEFORE their first event so that the reported kernel time is the kernel's
 * (cuda/cuda_wm.cu:271-283 brackets the launch only) */
static int wm_prepare(struct smh_wm *wm, int variant);
static int wm_ensure_device(struct smh_wm *wm, smh_wm_dev **out);
/* the automaton kernels' verify stage hashes the window when the handle carries a verify table (ac_host.c hv_wm) */
static int ac_fill_cold(struct smh_ac *ac, smh_ac_dev *dv, smh_ac_verify_ctx &V)
{
    {
        std::lock_guard<std::mutex> lock(g_dev_mu);
        if (dv->d_cold) { V.cold = dv->d_cold; return SMH_OK; }
    }
    smh_ac_cold_ctx C = {};
    C.full = dv->d_table; C.full_entry_bytes = ac->entry_bytes; C.depth_first = dv->d_depth_first;
    C.trunc1 = dv->d_trunc1; C.trunc1_entry_bytes = ac->trunc1_entry_bytes;
    if (ac->hv_wm && ac->hv_wm->verify) {
        smh_wm_dev *hd = NULL;
        const int rc = wm_ensure_device(ac->hv_wm, &hd);
        if (rc != SMH_OK) return rc;
        C.hv_verify = hd->d_verify; C.hv_pats = hd->d_pat_sorted; C.hv_log2 = ac->hv_wm->verify_log2;
    }
    smh_ac_cold_ctx *d = NULL;
    HIP_TRY(hipMalloc((void **)&d, sizeof C));
    if (hipMemcpy(d, &C, sizeof C, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d);
        smh_set_error("smh_ac_scan: upload of the verify context failed");
        return SMH_ENODEV;
    }
    {
        std::lock_guard<std::mutex> lock(g_dev_mu);
        if (!dv->d_cold) { dv->d_cold = d; d = NULL; }
    }
    if (d) (void)hipFree(d); /* another thread published one meanwhile */
    V.cold = dv->d_cold;
    return SMH_OK;
}
static int ac_engine_static(const struct smh_ac *ac)
{
    if (ac->engine_forced >= 0) return ac->engine_forced;
    return ac->alt_wm ? SMH_ALGO_WM : SMH_ALGO_AC;
}
static struct smh_wm *ac_filter_engine(const struct smh_ac *ac) { return ac->alt_wm ? ac->alt_wm : ac->flex_wm; }
/* the engine the next tuned scan on the current device runs (positions, info) */
/* a handle with per-device state shared by its launches: several engines (reports, rates) or a candidate-queue workspace */
static struct smh_hashes *ac_hashes(const struct smh_ac *ac) { return smh_ac_hash_engine(ac); }
static bool ac_needs_state(const struct smh_ac *ac) { return ac->flex_wm || ac->flat_ac || ac->keys || ac_hashes(ac) || (!ac->scan_exact && !ac->scan_dense); }
static bool ac_adaptive(const struct smh_ac *ac) { return (ac->flex_wm || ac->flat_ac || ac->keys || ac_hashes(ac)) && adapt_enabled(); }
/* the text-independent engine: one exact stride-1 launch per part (ac_host.c, end of the compile) */
static double ac_flat_ms(const struct smh_ac *ac)
{
    double ms = 0.0;
    for (const struct smh_ac *p = ac->flat_ac; p; p = p->flat_next) ms += smh_ac_plan_ms(p);
    /* the plan model prices an exact stride-1 scan at 0.289 ms/GiB; the parts measure 0.25 each, a set's parts together 0.76-0.90
     * of the sum (tools/est_check.py on 4- and 20-letter sets of 1 to 11 parts: profiles/r04_final/notes/est_check_*.log) */
    return 0.87 * ms;
}
static void ac_estimates(const struct smh_ac *ac, double est[SMH_ENGINES])
{
    /* a verify-bound plan (the compile handed the set to the filter kernels for that reason) is no candidate: on the texts
     * that slow the filter down it is slower still (8000 patterns: 18-45 ms/GiB against 2.6-6.2, bench "skewed") */
    est[SMH_ALGO_AC] = ac->alt_wm && ac->scan_cost > SMH_AC_ALT_ENGINE_COST ? 0.0 : smh_ac_plan_ms(ac);
    est[SMH_ALGO_WM] = ac_filter_engine(ac) ? ac_filter_engine(ac)->scan_ms_est : 0.0;
    est[SMH_ENGINE_AC_FLAT] = ac_flat_ms(ac);
    est[SMH_ENGINE_KEYS] = ac->keys ? ac->keys->ms_est : 0.0;
    /* the window-hash engine of the handle's filter engine (byte-like alphabets), unless the handle has a key table of its own */
    est[SMH_ENGINE_HASH] = ac_hashes(ac) ? ac_hashes(ac)->ms_est : 0.0;
}
static int ac_engine_now(struct smh_ac *ac)
{
    if (ac->engine_forced < 0 && ac_adaptive(ac))
        if (smh_adapt_dev *A = adapt_find(&ac->adapt)) {
            std::lock_guard<std::mutex> lock(*A->mu);
            if (A->engine >= 0) return A->engine;
        }
    return ac_engine_static(ac);
}

static int ac_prepare(struct smh_ac *ac, int variant);
static int ac_flat_prepare(struct smh_ac *ac)
{
    int rc = SMH_OK;
    for (struct smh_ac *p = ac->flat_ac; p && rc == SMH_OK; p = p->flat_next) rc = ac_prepare(p, SMH_VARIANT_TUNED);
    return rc;
}

static int ac_prepare_engines(struct smh_ac *ac, int variant)
{
    const bool both = variant == SMH_VARIANT_TUNED && ac->engine_forced < 0 && ac_adaptive(ac);
    if (variant == SMH_VARIANT_TUNED && ac_engine_static(ac) == SMH_ALGO_WM && !both) return wm_prepare(ac_filter_engine(ac), variant);
    if (variant == SMH_VARIANT_TUNED && ac_engine_static(ac) == SMH_ENGINE_AC_FLAT && !both) return ac_flat_prepare(ac);
    if (variant == SMH_VARIANT_TUNED && ac_engine_static(ac) == SMH_ENGINE_KEYS && !both && ac->keys) { smh_keys_dev *kd = NULL; return keys_ensure_device(ac->keys, &kd); }
    if (variant == SMH_VARIANT_TUNED && ac_engine_static(ac) == SMH_ENGINE_HASH && !both && ac_hashes(ac)) { smh_hash_dev *hd = NULL; return hash_ensure_device(ac_hashes(ac), &hd); }
    smh_ac_dev *d = NULL;
    int rc = ac_ensure_device(ac, &d);
    if (rc == SMH_OK && variant == SMH_VARIANT_TABLE) rc = ac_ensure_reference_tables(ac, d);
    if (rc == SMH_OK && variant == SMH_VARIANT_TUNED) {
        smh_ac_verify_ctx V = {};
        rc = ac_fill_cold(ac, d, V); /* the hash-verify handle's tables and the kernels' cold context */
    }
    if (rc == SMH_OK && both) { /* any of the engines may serve the next launch */
        if (ac_filter_engine(ac)) rc = wm_prepare(ac_filter_engine(ac), variant);
        if (rc == SMH_OK) rc = ac_flat_prepare(ac);
        if (rc == SMH_OK && ac->keys) { smh_keys_dev *kd = NULL; rc = keys_ensure_device(ac->keys, &kd); }
        if (rc == SMH_OK && ac_hashes(ac)) { smh_hash_dev *hd = NULL; rc = hash_ensure_device(ac_hashes(ac), &hd); }
    }
    return rc;
}
static int ac_prepare(struct smh_ac *ac, int variant)
{
    int rc = ac_prepare_engines(ac, variant);
    /* the adaptive state too, whenever smh_ac_scan will ask for it (also with an engine forced): created inside the scan call it
     * would put blocking allocations into a captured stream and into count_host's kernel time */
    if (rc == SMH_OK && variant == SMH_VARIANT_TUNED && ac_needs_state(ac)) {
        smh_adapt_dev *A = NULL;
        rc = adapt_get(&ac->adapt, &A);
    }
    return rc;
}

/* everything the tuned scans of a handle need on the CURRENT device, without launching (smh_multi.hip prepares
 * every device of a node from a thread of its own before the first timed scan) */
extern "C" int smh_ac_prepare_device(struct smh_ac *ac)
{
    if (!ac || ac->magic != SMH_MAGIC_AC) { smh_set_error("smh_ac_prepare_device: bad handle"); return SMH_EINVAL; }
    return ac_prepare(ac, SMH_VARIANT_TUNED);
}
extern "C" int smh_wm_prepare_device(struct smh_wm *wm)
{
    if (!wm || wm->magic != SMH_MAGIC_WM) { smh_set_error("smh_wm_prepare_device: bad handle"); return SMH_EINVAL; }
    return wm_prepare(wm, SMH_VARIANT_TUNED);
}

static int wm_launch_own(struct smh_wm *wm, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream,
                         const smh_stats_arg &SA, float density);

/* the automaton kernels on the plan the handle holds (no engine routing) */
static int ac_launch_own(struct smh_ac *ac, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream,
                         const smh_stats_arg &SA)
{
    smh_ac_dev *dv = NULL;
    int rc = ac_ensure_device(ac, &dv);
    if (rc != SMH_OK) return rc;
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    if (ac->scan_dense) {
        /* the dense plan: the rolling code of the last m symbols IS the state; two END columns per lookup of the accepting-
         * set bits (the lane code of wm_pair_kernel: smh_wm_pair_thread in wm_lane.h, here fed with the automaton's bits) */
        smh_wm_launch W = {};
        W.d_text = d_text; W.n = n; W.m = ac->m; W.bits = 2; W.block_symbols = ac->m; W.filter_log2 = 2 * ac->m; W.filter_exact = 1;
        W.d_filter = dv->d_dense_filter; W.d_pair = dv->d_dense_pair; W.d_count = d_count; W.n_cus = n_cus;
        HIP_TRY(smh_launch_wm_block(W, (hipStream_t)stream));
        return SMH_OK;
    }
    smh_ac_launch L = {};
    L.V.text = d_text; L.V.n = n; L.V.m = ac->m; L.V.K = ac->scan_depth; L.V.sigma = ac->alphabet;
    if ((rc = ac_fill_cold(ac, dv, L.V)) != SMH_OK) return rc;
    L.stride = ac->scan_stride; L.exact = ac->scan_exact; L.scan_entry_bytes = ac->scan_entry_bytes;
    L.d_scan_table = dv->d_scan; L.lds_bytes = ac->scan_bytes; L.d_queue = dv->d_queue;
    for (int i = 0; i < SMH_AC_DF_LEN; ++i) L.df.v[i] = i <= ac->max_depth + 1 ? ac->depth_first[i] : ac->rows;
    L.full_rows = ac->scan_full_rows;
    if (L.full_rows) /* hybrid image: compact rows are not numbered by depth; the halo's "deep enough"
                      * test treats every compact row as deep (conservative, see ac_lane.h) */
        for (int i = 0; i < SMH_AC_DF_LEN; ++i)
            if (L.df.v[i] > L.full_rows) L.df.v[i] = L.full_rows;
    /* development knob SMH_AC_TUNE="nohalo=1": no lane ever counts as deep enough for a halo step, so the scan does NONE of its
     * K - 1 warm-up steps -- counts are wrong; the launch time is the bound on what shrinking the halo share could buy
     * (profiles/r05_final/notes/ab_automaton_halo.log) */
    if (smh_tune_has(SMH_TUNE_AC, "nohalo=1")) /* testing library only (smh_tune.h) */
        for (int i = 0; i < SMH_AC_DF_LEN; ++i) L.df.v[i] = 0xFFFFFFFFu;
    L.d_count = d_count; L.n_cus = n_cus;
    L.V.pos.out = NULL; L.V.pos.capacity = 0; L.V.pos.cursor = NULL;
    L.d_wave_times = g_wave_trace;
    L.stats = SA;
    HIP_TRY(smh_launch_ac_dfa(L, (hipStream_t)stream));
    return SMH_OK;
}

/* the parts of the text-independent engine one after the other into the same count; the first launch reports for all
 * of them (its duration times the number of parts: a plain stride-1 scan runs at one speed whatever its table) */
static int ac_flat_launch(struct smh_ac *ac, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream, smh_stats_arg SA)
{
    if (SA.st) SA.tag |= (unsigned int)(ac->flat_parts > 0 ? ac->flat_parts : 1) << 8;
    for (struct smh_ac *p = ac->flat_ac; p; p = p->flat_next) {
        const int rc = ac_launch_own(p, d_text, n, d_count, stream, SA);
        if (rc != SMH_OK) return rc;
        SA = smh_stats_arg{};
    }
    return SMH_OK;
}

/* the parts append to one output: distinct patterns of one length never share an END column */
static int ac_flat_positions(struct smh_ac *ac, const unsigned char *d_text, uint64_t n, uint64_t *d_positions, uint64_t capacity,
                             uint64_t *d_cursor, void *stream)
{
    for (struct smh_ac *p = ac->flat_ac; p; p = p->flat_next)
        if (const int rc = smh_ac_positions(p, d_text, n, d_positions, capacity, d_cursor, stream); rc != SMH_OK) return rc;
    return SMH_OK;
}

extern "C" int smh_ac_scan(smh_ac *ac, const unsigned char *d_text, uint64_t n, uint64_t *d_count, int variant,
                           void *stream)
{
    if (!ac || ac->magic != SMH_MAGIC_AC || !d_count || (n && !d_text)) {
        smh_set_error("smh_ac_scan: bad arguments");
        return SMH_EINVAL;
    }
    if (((uintptr_t)d_text & 15u) != 0) {
        smh_set_error("smh_ac_scan: d_text must be 16-byte aligned");
        return SMH_EINVAL;
    }
    if (!ac->fixed_length_ok) {
        smh_set_error("smh_ac_scan: the automaton's accepting states are not all leaves at depth m = %d; "
                      "the segmented scan is exact only for patterns of one length (as is the reference's "
                      "GPU path, cuda/cuda_ac.cu:31-34)", ac->m);
        return SMH_EUNSUP;
    }
    if (n < (uint64_t)ac->m) return SMH_OK;
    if (variant == SMH_VARIANT_TUNED) {
        /* engine choice: ac_host.c, end of the compile; with both engines at hand, what the launches report (above) */
        int engine = ac_engine_static(ac), rc;
        auto launch_static = [&]() -> int {
            if (engine == SMH_ALGO_WM) return smh_wm_scan(ac_filter_engine(ac), d_text, n, d_count, SMH_VARIANT_TUNED, stream);
            if (engine == SMH_ENGINE_AC_FLAT && ac->flat_ac) return ac_flat_launch(ac, d_text, n, d_count, stream, smh_stats_arg{});
            if (engine == SMH_ENGINE_KEYS && ac->keys) return keys_launch(ac->keys, d_text, n, d_count, NULL, stream, smh_stats_arg{});
            if (engine == SMH_ENGINE_HASH && ac_hashes(ac)) return hash_launch(ac_hashes(ac), d_text, n, d_count, NULL, stream, smh_stats_arg{});
            return ac_launch_own(ac, d_text, n, d_count, stream, smh_stats_arg{});
        };
        if (!ac_needs_state(ac)) return launch_static(); /* one exact plan, no workspace, nothing to adapt: launches may overlap freely */
        smh_adapt_dev *A = NULL;
        if ((rc = adapt_get(&ac->adapt, &A)) != SMH_OK) return rc;
        std::lock_guard<std::mutex> adapt_lock(*A->mu); /* poll, choice, slot and launch of one handle happen one at a time (smh_stats.h SMH_STATS_SLOTS) */
        if ((rc = adapt_order_before(A, stream)) != SMH_OK) return rc;
        rc = [&]() -> int {
            int rc;
            if (!ac_adaptive(ac)) return launch_static();
            adapt_poll(A);
            if (ac->engine_forced < 0) {
                double est[SMH_ENGINES];
                ac_estimates(ac, est);
                uint64_t done = 0;
                rc = adapt_first_look(A, est, engine, ac->m, d_text, n, d_count, stream,
                                      [&](int e, const unsigned char *t, uint64_t len, uint64_t *cnt, const smh_stats_arg &sa) -> int {
                                          if (e == SMH_ALGO_WM) return wm_launch_own(ac_filter_engine(ac), t, len, cnt, stream, sa, adapt_density(A, ac_filter_engine(ac)));
                                          if (e == SMH_ENGINE_AC_FLAT) return ac_flat_launch(ac, t, len, cnt, stream, sa);
                                          if (e == SMH_ENGINE_KEYS) return keys_launch(ac->keys, t, len, cnt, NULL, stream, sa);
                                          if (e == SMH_ENGINE_HASH) return hash_launch(ac_hashes(ac), t, len, cnt, NULL, stream, sa);
                                          return ac_launch_own(ac, t, len, cnt, stream, sa);
                                      }, &done);
                if (rc != SMH_OK) return rc;
                d_text += done; /* a multiple of 16 */
                n -= done;
                engine = adapt_choose(A, est, engine);
            } else {
                A->engine = engine;
            }
            /* (short texts: the key image instead of a big-table filter, as in smh_wm_scan below) */
            if (engine == SMH_ALGO_WM && ac->engine_forced < 0 && ac->keys && n < SMH_ADAPT_MIN_BYTES &&
                (ac_filter_engine(ac)->gram_kind == SMH_GRAM_BYTE_BIG || ac_filter_engine(ac)->gram_kind == SMH_GRAM_FLAT_BIG || ac_filter_engine(ac)->gram_kind == SMH_GRAM_FLAT4_BIG))
                return keys_launch(ac->keys, d_text, n, d_count, NULL, stream, adapt_arg(A, n, SMH_ENGINE_KEYS, stream));
            const smh_stats_arg SA = adapt_arg(A, n, engine, stream);
            if (engine == SMH_ALGO_WM) return wm_launch_own(ac_filter_engine(ac), d_text, n, d_count, stream, SA, adapt_density(A, ac_filter_engine(ac)));
            if (engine == SMH_ENGINE_AC_FLAT) return ac_flat_launch(ac, d_text, n, d_count, stream, SA);
            if (engine == SMH_ENGINE_KEYS) return keys_launch(ac->keys, d_text, n, d_count, NULL, stream, SA);
            if (engine == SMH_ENGINE_HASH) return hash_launch(ac_hashes(ac), d_text, n, d_count, NULL, stream, SA);
            return ac_launch_own(ac, d_text, n, d_count, stream, SA);
        }();
        const int rc_after = adapt_order_after(A, stream);
        return rc != SMH_OK ? rc : rc_after;
    }
    if (variant != SMH_VARIANT_TABLE) {
        smh_set_error("smh_ac_scan: unknown variant %d", variant);
        return SMH_EINVAL;
    }
    smh_ac_dev *dv = NULL;
    int rc = ac_ensure_device(ac, &dv);
    if (rc != SMH_OK) return rc;
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    if ((rc = ac_ensure_reference_tables(ac, dv)) != SMH_OK) return rc;
    smh_ac_table_launch L;
    L.d_text = d_text; L.n = n; L.m = ac->m; L.alphabet = ac->alphabet;
    L.d_transition = dv->d_transition; L.d_supply = dv->d_supply; L.d_final = dv->d_final;
    L.d_count = d_count; L.n_cus = n_cus;
    HIP_TRY(smh_launch_ac_table(L, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_ac_get_adapt(smh_ac *ac, smh_adapt_info *out)
{
    if (!ac || ac->magic != SMH_MAGIC_AC || !out || out->struct_size != sizeof *out) {
        smh_set_error("smh_ac_get_adapt: bad arguments (set struct_size = sizeof(smh_adapt_info))");
        return SMH_EINVAL;
    }
    double est[SMH_ENGINES];
    ac_estimates(ac, est);
    if (!ac->flex_wm && ac->alt_wm) est[SMH_ALGO_WM] = ac->alt_wm->scan_ms_est;
    const int adaptive = ac->engine_forced < 0 && ac_adaptive(ac);
    smh_adapt_dev *A = ac_adaptive(ac) ? adapt_find(&ac->adapt) : NULL;
    std::unique_lock<std::mutex> lock;
    if (A) lock = std::unique_lock<std::mutex>(*A->mu);
    if (A) adapt_poll(A); /* what the launches finished so far have published */
    adapt_report(A, adaptive, ac_engine_static(ac), est, out);
    return SMH_OK;
}

static int ac_positions_impl(smh_ac *ac, int engine, const unsigned char *d_text, uint64_t n, uint64_t *d_positions, uint64_t capacity,
                             uint64_t *d_cursor, void *stream);
extern "C" int smh_ac_positions(smh_ac *ac, const unsigned char *d_text, uint64_t n, uint64_t *d_positions,
                                uint64_t capacity, uint64_t *d_cursor, void *stream)
{
    if (!ac || ac->magic != SMH_MAGIC_AC || !d_cursor || (capacity && !d_positions) || (n && !d_text)) {
        smh_set_error("smh_ac_positions: bad arguments");
        return SMH_EINVAL;
    }
    if (!ac_needs_state(ac)) return ac_positions_impl(ac, ac_engine_static(ac), d_text, n, d_positions, capacity, d_cursor, stream);
    /* ordered behind the handle's other launches on this device (adapt_order_before): the depth-cut kernels' queue workspace is shared */
    smh_adapt_dev *A = NULL;
    int rc = adapt_get(&ac->adapt, &A);
    if (rc != SMH_OK) return rc;
    const int engine = ac_engine_now(ac); /* (takes A->mu itself) */
    std::lock_guard<std::mutex> adapt_lock(*A->mu);
    if ((rc = adapt_order_before(A, stream)) != SMH_OK) return rc;
    rc = ac_positions_impl(ac, engine, d_text, n, d_positions, capacity, d_cursor, stream);
    const int rc_after = adapt_order_after(A, stream);
    return rc != SMH_OK ? rc : rc_after;
}
static int ac_positions_impl(smh_ac *ac, int engine, const unsigned char *d_text, uint64_t n, uint64_t *d_positions, uint64_t capacity,
                             uint64_t *d_cursor, void *stream)
{
    if (!ac->fixed_length_ok) {
        smh_set_error("smh_ac_positions: patterns are not all of length m");
        return SMH_EUNSUP;
    }
    if (n < (uint64_t)ac->m) return SMH_OK;
    if (engine == SMH_ALGO_WM)
        return smh_wm_positions(ac_filter_engine(ac), d_text, n, d_positions, capacity, d_cursor, stream);
    else if (engine == SMH_ENGINE_AC_FLAT)
        return ac_flat_positions(ac, d_text, n, d_positions, capacity, d_cursor, stream);
    else if (engine == SMH_ENGINE_KEYS && ac->keys && ((uintptr_t)d_text & 15u) == 0) {
        const smh_pos_out po = {d_positions, capacity, d_cursor};
        return keys_launch(ac->keys, d_text, n, NULL, &po, stream, smh_stats_arg{});
    } else if (engine == SMH_ENGINE_HASH && ac_hashes(ac) && ((uintptr_t)d_text & 15u) == 0) {
        const smh_pos_out po = {d_positions, capacity, d_cursor};
        return hash_launch(ac_hashes(ac), d_text, n, NULL, &po, stream, smh_stats_arg{});
    }
    smh_ac_dev *dv = NULL;
    int rc = ac_ensure_device(ac, &dv);
    if (rc != SMH_OK) return rc;
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    if (ac->scan_dense && ((uintptr_t)d_text & 15u) == 0) {
        smh_wm_launch W = {};
        W.d_text = d_text; W.n = n; W.m = ac->m; W.bits = 2; W.block_symbols = ac->m; W.filter_log2 = 2 * ac->m; W.filter_exact = 1;
        W.d_filter = dv->d_dense_filter; W.d_pair = dv->d_dense_pair; W.d_count = NULL; W.n_cus = n_cus;
        W.po.out = d_positions; W.po.capacity = capacity; W.po.cursor = d_cursor;
        HIP_TRY(smh_launch_wm_block_positions(W, (hipStream_t)stream));
        return SMH_OK;
    }
    smh_ac_verify_ctx V = {};
    V.text = d_text; V.n = n; V.m = ac->m; V.K = ac->scan_depth; V.sigma = ac->alphabet;
    if ((rc = ac_fill_cold(ac, dv, V)) != SMH_OK) return rc;
    /* the tuned scan kernels in positions mode: matches are recorded as bits and appended per wave */
    smh_ac_launch L = {};
    L.V = V;
    L.stride = ac->scan_stride; L.exact = ac->scan_exact; L.scan_entry_bytes = ac->scan_entry_bytes;
    L.d_scan_table = dv->d_scan; L.lds_bytes = ac->scan_bytes; L.d_queue = dv->d_queue;
    for (int i = 0; i < SMH_AC_DF_LEN; ++i) L.df.v[i] = i <= ac->max_depth + 1 ? ac->depth_first[i] : ac->rows;
    L.full_rows = ac->scan_full_rows;
    if (L.full_rows)
        for (int i = 0; i < SMH_AC_DF_LEN; ++i)
            if (L.df.v[i] > L.full_rows) L.df.v[i] = L.full_rows;
    L.d_count = NULL; L.n_cus = n_cus;
    L.V.pos.out = d_positions; L.V.pos.capacity = capacity; L.V.pos.cursor = d_cursor;
    if (((uintptr_t)d_text & 15u) == 0) {
        const hipError_t e = smh_launch_ac_dfa_positions(L, (hipStream_t)stream);
        if (e == hipSuccess) return SMH_OK;
        if (e != hipErrorNotSupported) HIP_TRY(e);
    }
    /* plans with a halo beyond 32 bytes, unaligned text: one lane per segment over the stride-1 table in HBM */
    HIP_TRY(smh_launch_ac_positions(V, d_positions, capacity, d_cursor, n_cus, (hipStream_t)stream));
    return SMH_OK;
}

/* ---- the legacy host-pointer path: what search_ac / search_wu / cuda_* run (the text is a pageable host buffer, as
 * main.c:582-648 hands it over).  Round 3 did hipMalloc(n) + one copy + kernel + hipFree per call: 36 GB/s.  Now the text
 * crosses PCIe in pieces of SMH_HOST_PIECE bytes through TWO device buffers of a grow-only workspace: piece k+1 is
 * copied (copy stream) while piece k is scanned (scan stream), so device memory is 2 x (piece + halo) whatever n is and
 * nothing is allocated per call.  The piece math is the shard formula (main.c:467-477, smh_shard_range): piece k holds
 * text[k P, min((k+1) P + m - 1, n)) and its scan counts the END columns that lie in it with a whole window -- every END
 * column of [m-1, n) belongs to exactly one piece.  Workspaces are pooled per device and handed to one call at a time;
 * nothing is cached by the caller's pointer (the caller may rewrite its buffer between calls).
 * SMH_HOST_PIECE_KIB (environment) overrides the piece size: the tests run it at 4 KiB so that a small text has hundreds
 * of piece boundaries. */
#define SMH_HOST_PIECE (64ull << 20)
struct smh_host_ws {
    int device;
    unsigned char *d_buf[2];
    uint64_t cap; /* bytes of each buffer */
    uint64_t *d_count;
    hipStream_t copy_stream, scan_stream;
    hipEvent_t copied[2], k0[2], k1[2];
    smh_host_ws *next;
};
static std::mutex g_ws_mu;
static smh_host_ws *g_ws_free = NULL;

static void host_ws_destroy(smh_host_ws *w)
{
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(w->d_buf[b]);
        if (w->copied[b]) (void)hipEventDestroy(w->copied[b]);
        if (w->k0[b]) (void)hipEventDestroy(w->k0[b]);
        if (w->k1[b]) (void)hipEventDestroy(w->k1[b]);
    }
    (void)hipFree(w->d_count);
    if (w->copy_stream) (void)hipStreamDestroy(w->copy_stream);
    if (w->scan_stream) (void)hipStreamDestroy(w->scan_stream);
    delete w;
}

/* frees the pooled workspaces of every device (they are kept between calls otherwise); for leak checks and tidy exits */
static void legacy_release(void); /* the handles the legacy GPU names keep between calls (end of this file) */
extern "C" void smh_host_path_release(void)
{
    legacy_release();
    smh_host_ws *w;
    {
        std::lock_guard<std::mutex> lock(g_ws_mu);
        w = g_ws_free;
        g_ws_free = NULL;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    while (w) {
        smh_host_ws *next = w->next;
        if (hipSetDevice(w->device) == hipSuccess) host_ws_destroy(w);
        w = next;
    }
    (void)hipSetDevice(prev);
}

static int host_ws_acquire(uint64_t need, smh_host_ws **out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    smh_host_ws *w = NULL;
    {
        std::lock_guard<std::mutex> lock(g_ws_mu);
        for (smh_host_ws **pp = &g_ws_free; *pp; pp = &(*pp)->next)
            if ((*pp)->device == dev) { w = *pp; *pp = w->next; break; }
    }
    if (!w) {
        w = new smh_host_ws();
        memset(w, 0, sizeof *w);
        w->device = dev;
        hipError_t e = hipStreamCreateWithFlags(&w->copy_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&w->scan_stream, hipStreamNonBlocking);
        for (int b = 0; b < 2 && e == hipSuccess; ++b) {
            e = hipEventCreateWithFlags(&w->copied[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreate(&w->k0[b]);
            if (e == hipSuccess) e = hipEventCreate(&w->k1[b]);
        }
        if (e == hipSuccess) e = hipMalloc((void **)&w->d_count, 16);
        if (e != hipSuccess) {
            smh_set_error("host-pointer path: workspace: %s", hipGetErrorString(e));
            host_ws_destroy(w);
            return SMH_ENODEV;
        }
    }
    if (w->cap < need) { /* grow-only */
        for (int b = 0; b < 2; ++b) { (void)hipFree(w->d_buf[b]); w->d_buf[b] = NULL; }
        w->cap = 0;
        for (int b = 0; b < 2; ++b) {
            const hipError_t e = hipMalloc((void **)&w->d_buf[b], need);
            if (e != hipSuccess) {
                smh_set_error("host-pointer path: hipMalloc(%llu): %s", (unsigned long long)need, hipGetErrorString(e));
                host_ws_destroy(w);
                return e == hipErrorOutOfMemory ? SMH_ENOMEM : SMH_ENODEV;
            }
        }
        w->cap = need;
    }
    *out = w;
    return SMH_OK;
}

static void host_ws_release(smh_host_ws *w)
{
    std::lock_guard<std::mutex> lock(g_ws_mu);
    w->next = g_ws_free;
    g_ws_free = w;
}

/* piece size of the host-pointer path: SMH_HOST_PIECE, or SMH_HOST_PIECE_KIB from the environment READ ONCE per process, or what
 * smh_host_path_set_piece() last set.  It moves no count: pieces overlap by m - 1 bytes whatever their size. */
static std::atomic<uint64_t> g_host_piece{0};

static uint64_t host_piece_bytes(void)
{
    uint64_t p = g_host_piece.load(std::memory_order_relaxed);
    if (p) return p;
    static const uint64_t from_env = [] {
        uint64_t v = SMH_HOST_PIECE;
        if (const char *e = getenv("SMH_HOST_PIECE_KIB")) {
            const long long kib = atoll(e);
            if (kib >= 4) v = ((uint64_t)kib << 10) & ~(uint64_t)4095;
        }
        return v;
    }();
    return from_env;
}

extern "C" uint64_t smh_host_path_set_piece(uint64_t bytes)
{
    const uint64_t before = host_piece_bytes();
    if (bytes == 0) g_host_piece.store(0, std::memory_order_relaxed); /* back to the default / the environment's value */
    else g_host_piece.store(bytes < 4096 ? 4096 : bytes & ~(uint64_t)4095, std::memory_order_relaxed);
    return before;
}

/* shared by the *_count_host helpers.  m = the pattern length (pieces overlap by m - 1 bytes); prepare() = table uploads,
 * outside every event: *kernel_seconds is the kernels' time alone, as the reference's cudaEvents bracket the launch only
 * (cuda/cuda_wm.cu:271-283); launch(d_text, len, d_count, stream) = one asynchronous scan */
template <typename Prepare, typename Launch>
static int count_host(const unsigned char *text, uint64_t n, int m, uint64_t *count, double *kernel_seconds, Prepare prepare,
                      Launch launch)
{
    if (!count || (n && !text)) { smh_set_error("count_host: bad arguments"); return SMH_EINVAL; }
    *count = 0;
    if (kernel_seconds) *kernel_seconds = 0.0;
    if (m < 1) m = 1;
    uint64_t P = host_piece_bytes();
    const uint64_t halo = (uint64_t)(m - 1);
    if (halo * 8u > P || n <= P + halo) P = n > 4096u ? (n + 4095u) & ~(uint64_t)4095 : 4096u; /* one piece */
    smh_host_ws *w = NULL;
    int rc = host_ws_acquire(P + ((halo + 15u) & ~(uint64_t)15) + 4096u, &w);
    if (rc != SMH_OK) return rc;
    double ksecs = 0.0;
    uint64_t n_pieces = 0;
#define CH_TRY(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            smh_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            rc = SMH_ENODEV;                                                               \
            goto done;                                                                     \
        }                                                                                  \
    } while (0)
    CH_TRY(hipMemsetAsync(w->d_count, 0, 16, w->scan_stream));
    rc = prepare();
    if (rc != SMH_OK) goto done;
    for (uint64_t off = 0; off < n; off += P, ++n_pieces) {
        const int b = (int)(n_pieces & 1u);
        uint64_t len = n - off;
        if (len > P + halo) len = P + halo;
        if (len < (uint64_t)m && n_pieces) break; /* the tail inside the piece before's halo: no END column of its own */
        if (n_pieces >= 2) { /* the buffer's last scan must be over before it is overwritten */
            CH_TRY(hipEventSynchronize(w->k1[b]));
            float ms = 0.f;
            if (kernel_seconds) { CH_TRY(hipEventElapsedTime(&ms, w->k0[b], w->k1[b])); ksecs += (double)ms / 1000.0; }
        }
        CH_TRY(hipMemcpyAsync(w->d_buf[b], text + off, len, hipMemcpyHostToDevice, w->copy_stream));
        CH_TRY(hipEventRecord(w->copied[b], w->copy_stream));
        CH_TRY(hipStreamWaitEvent(w->scan_stream, w->copied[b], 0));
        CH_TRY(hipEventRecord(w->k0[b], w->scan_stream));
        rc = launch(w->d_buf[b], len, w->d_count, (void *)w->scan_stream);
        if (rc != SMH_OK) goto done;
        CH_TRY(hipEventRecord(w->k1[b], w->scan_stream));
    }
    CH_TRY(hipMemcpyAsync(count, w->d_count, 8, hipMemcpyDeviceToHost, w->scan_stream));
    CH_TRY(hipStreamSynchronize(w->scan_stream));
    if (kernel_seconds) {
        for (uint64_t k = n_pieces >= 2 ? n_pieces - 2 : 0; k < n_pieces; ++k) {
            float ms = 0.f;
            CH_TRY(hipEventElapsedTime(&ms, w->k0[k & 1u], w->k1[k & 1u]));
            ksecs += (double)ms / 1000.0;
        }
        *kernel_seconds = ksecs;
    }
done:
    if (rc != SMH_OK) { /* leave nothing in flight on a workspace that goes back to the pool */
        (void)hipStreamSynchronize(w->copy_stream);
        (void)hipStreamSynchronize(w->scan_stream);
    }
    host_ws_release(w);
    return rc;
#undef CH_TRY
}

extern "C" int smh_ac_count_host(smh_ac *ac, const unsigned char *text, uint64_t n, int variant, uint64_t *count,
                                 double *kernel_seconds)
{
    if (!ac || ac->magic != SMH_MAGIC_AC) { smh_set_error("smh_ac_count_host: bad handle"); return SMH_EINVAL; }
    return count_host(text, n, ac->m, count, kernel_seconds,
                      [&]() { return n < (uint64_t)ac->m || !ac->fixed_length_ok ? SMH_OK : ac_prepare(ac, variant); },
                      [&](unsigned char *d_text, uint64_t len, uint64_t *d_count, void *st) { return smh_ac_scan(ac, d_text, len, d_count, variant, st); });
}

/* ------------------------------------------------------------------ WM */
static void wm_dev_free_one(smh_wm_dev *dev)
{
    (void)hipFree(dev->d_filter);
    (void)hipFree(dev->d_pair);
    (void)hipFree(dev->d_gram);
    (void)hipFree(dev->d_queue);
    (void)hipFree(dev->d_verify);
    (void)hipFree(dev->d_verify_ck);
    (void)hipFree(dev->d_pat_sorted);
    (void)hipFree(dev->d_shift);
    (void)hipFree(dev->d_bucket_off);
    (void)hipFree(dev->d_bucket);
    (void)hipFree(dev->d_pat_orig);
    (void)hipFree(dev->d_classes);
    delete dev;
}

extern "C" void smh_wm_dev_free(struct smh_wm_dev *dev) /* the whole list */
{
    while (dev) {
        smh_wm_dev *next = dev->next;
        wm_dev_free_one(dev);
        dev = next;
    }
}

static int wm_ensure_device(struct smh_wm *wm, smh_wm_dev **out)
{
    return ensure_device_set<smh_wm_dev>(&wm->dev, wm_dev_free_one, [&](smh_wm_dev *d) -> int {
        int rc;
        const size_t fbytes = ((size_t)1 << wm->filter_log2) / 8;
        if ((rc = upload((void **)&d->d_filter, wm->filter, fbytes, 0)) != SMH_OK) return rc;
        if (wm->pair_table && (rc = upload((void **)&d->d_pair, wm->pair_table, 65536, 0)) != SMH_OK) return rc;
        if (wm->gram_table && (rc = upload((void **)&d->d_gram, wm->gram_table, wm->gram_bytes, 0)) != SMH_OK) return rc;
        if (wm->verify) {
            if ((rc = upload((void **)&d->d_verify, wm->verify, ((size_t)1 << wm->verify_log2) * 4, 0)) != SMH_OK) return rc;
            if (wm->verify_ck && (rc = upload((void **)&d->d_verify_ck, wm->verify_ck, (size_t)wm->ck_buckets * 16u, 16)) != SMH_OK) return rc;
        }
        {
            /* distinct patterns, each zero-padded to whole dwords (the verify stage compares dwords) */
            const size_t row = (size_t)((wm->m + 3) / 4) * 4;
            std::vector<unsigned char> padded((size_t)wm->distinct * row + 16, 0);
            for (int j = 0; j < wm->distinct; ++j) memcpy(padded.data() + (size_t)j * row, wm->pat_sorted + (size_t)j * wm->m, (size_t)wm->m);
            if ((rc = upload((void **)&d->d_pat_sorted, padded.data(), (size_t)wm->distinct * row, 16)) != SMH_OK) return rc;
        }
        /* room for a mixed-length set's class table (768 bytes; filled by wm_multi_launch when this handle is a set's suffix filter):
         * allocated with the set so that no scan ever allocates -- a scan may run inside a stream capture */
        HIP_TRY(hipMalloc((void **)&d->d_classes, sizeof(smh_wm_class) * SMH_WM_MAX_CLASSES));
        d->n_classes_up = 0;
        return SMH_OK;
    }, out);
}

/* the reference-layout tables (SHIFT, CSR buckets, patterns in the caller's order) go up on the first
 * table-walking scan only: the tuned kernels never read them */
static int wm_ensure_reference_tables(struct smh_wm *wm, smh_wm_dev *d)
{
    std::lock_guard<std::mutex> lock(g_dev_mu);
    if (d->d_pat_orig) return SMH_OK; /* the last to go up */
    int rc;
    std::vector<uint16_t> sh(wm->shiftsize);
    for (uint32_t i = 0; i < wm->shiftsize; ++i) {
        int32_t v = wm->l_shift[i];
        sh[i] = (uint16_t)(v < 0 ? 0 : (v > 65535 ? 65535 : v));
    }
    uint16_t *ds = NULL;
    uint32_t *dbo = NULL;
    int32_t *db = NULL;
    uint8_t *dp = NULL;
    if ((rc = upload((void **)&ds, sh.data(), sh.size() * 2, 0)) != SMH_OK ||
        (rc = upload((void **)&dbo, wm->l_bucket_off, ((size_t)wm->shiftsize + 1) * 4, 0)) != SMH_OK ||
        (rc = upload((void **)&db, wm->l_bucket, (size_t)wm->l_bucket_off[wm->shiftsize] * 8, 0)) != SMH_OK ||
        (rc = upload((void **)&dp, wm->pat_orig, (size_t)wm->patterns * wm->m, 0)) != SMH_OK) {
        (void)hipFree(ds); (void)hipFree(dbo); (void)hipFree(db); (void)hipFree(dp);
        return rc;
    }
    d->d_shift = ds; d->d_bucket_off = dbo; d->d_bucket = db; d->d_pat_orig = dp;
    return SMH_OK;
}

static int wm_engine_static(const struct smh_wm *wm)
{
    if (wm->engine_forced >= 0) return wm->engine_forced;
    return wm->alt_ac ? SMH_ALGO_AC : SMH_ALGO_WM;
}
static struct smh_ac *wm_automaton_engine(const struct smh_wm *wm) { return wm->alt_ac ? wm->alt_ac : wm->flex_ac; }
/* the automaton engine's estimate; 0 = kept for its text-independent parts only (wm_host.c, end of the compile) */
static double wm_flex_ms(const struct smh_wm *wm)
{
    if (!wm->flex_ac) return 0.0;
    return wm->alt_ac == wm->flex_ac || wm->flex_ac->scan_cost <= SMH_WM_FLEX_ENGINE_COST ? smh_ac_plan_ms(wm->flex_ac) : 0.0;
}
/* do this path's own kernels report (smh_stats.h)?  All but the pair lookup kernel (exact, m <= 8) */
static bool wm_reports(const struct smh_wm *wm) { return wm->gram_table || !wm->pair_table; }
static bool wm_engines(const struct smh_wm *wm) { return wm->flex_ac || wm->keys || wm->hashes; } /* more than this path's own kernels at hand */
static void wm_estimates(const struct smh_wm *wm, double est[SMH_ENGINES])
{
    est[SMH_ALGO_AC] = wm_flex_ms(wm);
    est[SMH_ALGO_WM] = wm->scan_ms_est;
    est[SMH_ENGINE_AC_FLAT] = wm->flex_ac ? ac_flat_ms(wm->flex_ac) : 0.0;
    est[SMH_ENGINE_KEYS] = wm->keys ? wm->keys->ms_est : 0.0;
    est[SMH_ENGINE_HASH] = wm->hashes ? wm->hashes->ms_est : 0.0;
}
static int wm_engine_now(struct smh_wm *wm)
{
    if (wm->engine_forced < 0 && wm_engines(wm) && adapt_enabled())
        if (smh_adapt_dev *A = adapt_find(&wm->adapt)) {
            std::lock_guard<std::mutex> lock(*A->mu);
            if (A->engine >= 0) return A->engine;
        }
    return wm_engine_static(wm);
}

static int wm_prepare(struct smh_wm *wm, int variant)
{
    const bool both = variant == SMH_VARIANT_TUNED && wm_engines(wm) && wm->engine_forced < 0 && adapt_enabled();
    const bool other = variant == SMH_VARIANT_TUNED && !both && wm_engine_static(wm) != SMH_ALGO_WM; /* a forced / static engine that is not this path's */
    int rc = SMH_OK;
    if (other && wm_engine_static(wm) == SMH_ALGO_AC) rc = ac_prepare(wm_automaton_engine(wm), variant);
    else if (other && wm_engine_static(wm) == SMH_ENGINE_AC_FLAT && wm->flex_ac) rc = ac_flat_prepare(wm->flex_ac);
    else if (other && wm_engine_static(wm) == SMH_ENGINE_KEYS && wm->keys) { smh_keys_dev *kd = NULL; rc = keys_ensure_device(wm->keys, &kd); }
    else if (other && wm_engine_static(wm) == SMH_ENGINE_HASH && wm->hashes) { smh_hash_dev *hd = NULL; rc = hash_ensure_device(wm->hashes, &hd); }
    if (other) {
        if (rc == SMH_OK && adapt_enabled() && (wm_engines(wm) || wm_reports(wm))) { /* smh_wm_scan asks for it whatever engine runs */
            smh_adapt_dev *A = NULL;
            rc = adapt_get(&wm->adapt, &A);
        }
        return rc;
    }
    smh_wm_dev *d = NULL;
    rc = wm_ensure_device(wm, &d);
    if (rc == SMH_OK && variant == SMH_VARIANT_TABLE) rc = wm_ensure_reference_tables(wm, d);
    if (rc == SMH_OK && both && wm->flex_ac) rc = ac_prepare(wm->flex_ac, variant);
    if (rc == SMH_OK && both && wm->keys) { smh_keys_dev *kd = NULL; rc = keys_ensure_device(wm->keys, &kd); }
    if (rc == SMH_OK && both && wm->hashes) { smh_hash_dev *hd = NULL; rc = hash_ensure_device(wm->hashes, &hd); }
    if (rc == SMH_OK && variant == SMH_VARIANT_TUNED && adapt_enabled() && (wm_engines(wm) || wm_reports(wm))) {
        smh_adapt_dev *A = NULL;
        rc = adapt_get(&wm->adapt, &A);
    }
    return rc;
}

/* this path's own tuned kernels (no engine routing); density = surviving columns per column the gram launcher plans for */
static int wm_launch_own(struct smh_wm *wm, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream,
                         const smh_stats_arg &SA, float density)
{
    smh_wm_dev *dv = NULL;
    int rc = wm_ensure_device(wm, &dv);
    if (rc != SMH_OK) return rc;
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    smh_wm_launch L = {};
    L.d_text = d_text; L.n = n; L.m = wm->m; L.bits = wm->bits_per_symbol; L.block_symbols = wm->block_symbols;
    L.filter_log2 = wm->filter_log2; L.filter_hashed = wm->filter_hashed; L.filter_k = wm->filter_k; L.filter_le4 = wm->filter_le4; L.filter_exact = wm->filter_exact;
    L.d_filter = dv->d_filter; L.d_pair = dv->d_pair; L.verify_log2 = wm->verify_log2; L.d_verify = dv->d_verify; L.d_verify_ck = dv->d_verify_ck; L.ck_buckets = wm->ck_buckets; L.ck_seed = wm->ck_seed;
    L.d_gram = dv->d_gram; L.gram_kind = wm->gram_kind; L.gram_density = density < 0 ? (float)wm->gram_density : density; L.gram_lane0 = (float)wm->gram_lane0; L.gram_planes = wm->gram_planes;
    if (wm->gram_kind == SMH_GRAM_FLAT || wm->gram_kind == SMH_GRAM_FLAT_BIG) L.gram_jb = wm->gram_jb; /* 1: two bits per gram */
    L.d_pat_sorted = dv->d_pat_sorted; L.d_queue = dv->d_queue; L.d_count = d_count; L.n_cus = n_cus;
    L.po.out = NULL; L.po.capacity = 0; L.po.cursor = NULL;
    L.stats = SA;
    HIP_TRY(smh_launch_wm_block(L, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_wm_scan(smh_wm *wm, const unsigned char *d_text, uint64_t n, uint64_t *d_count, int variant,
                           void *stream)
{
    if (!wm || wm->magic != SMH_MAGIC_WM || !d_count || (n && !d_text)) {
        smh_set_error("smh_wm_scan: bad arguments");
        return SMH_EINVAL;
    }
    if (((uintptr_t)d_text & 15u) != 0) {
        smh_set_error("smh_wm_scan: d_text must be 16-byte aligned");
        return SMH_EINVAL;
    }
    if (n < (uint64_t)wm->m) return SMH_OK;
    if (variant == SMH_VARIANT_TUNED) {
        /* engine choice: wm_host.c, end of the compile; with both engines at hand, what the launches report ("adaptive
         * engine" above).  A handle with one engine still learns its verify mode from its own reports. */
        int engine = wm_engine_static(wm), rc;
        smh_adapt_dev *A = NULL;
        std::unique_lock<std::mutex> adapt_lock; /* held to the end of the call: poll, choice, slot and launch of one handle one at a time */
        if (adapt_enabled() && (wm_engines(wm) || wm_reports(wm))) {
            if ((rc = adapt_get(&wm->adapt, &A)) != SMH_OK) return rc;
            adapt_lock = std::unique_lock<std::mutex>(*A->mu);
            if ((rc = adapt_order_before(A, stream)) != SMH_OK) return rc; /* launches of one handle: ordered on the device also across streams */
        }
        rc = [&]() -> int {
            int rc;
            if (A) {
                adapt_poll(A);
                if (wm_engines(wm) && wm->engine_forced < 0) {
                    double est[SMH_ENGINES];
                    wm_estimates(wm, est);
                    uint64_t done = 0;
                    rc = adapt_first_look(A, est, engine, wm->m, d_text, n, d_count, stream,
                                          [&](int e, const unsigned char *t, uint64_t len, uint64_t *cnt, const smh_stats_arg &sa) -> int {
                                              if (e == SMH_ALGO_AC) return ac_launch_own(wm_automaton_engine(wm), t, len, cnt, stream, sa);
                                              if (e == SMH_ENGINE_AC_FLAT) return ac_flat_launch(wm->flex_ac, t, len, cnt, stream, sa);
                                              if (e == SMH_ENGINE_KEYS) return keys_launch(wm->keys, t, len, cnt, NULL, stream, sa);
                                              if (e == SMH_ENGINE_HASH) return hash_launch(wm->hashes, t, len, cnt, NULL, stream, sa);
                                              return wm_launch_own(wm, t, len, cnt, stream, wm_reports(wm) ? sa : smh_stats_arg{}, adapt_density(A, wm));
                                          }, &done);
                    if (rc != SMH_OK) return rc;
                    d_text += done;
                    n -= done;
                    engine = adapt_choose(A, est, engine);
                } else {
                    A->engine = engine;
                }
            }
            /* Short texts (late round 6).  The engines are ranked by ms/GiB, but a launch over a few MiB is mostly its table staging:
             * 8000 protein patterns over the reference's 10.8 MB A.thaliana.faa take 24.8 us through the four-byte-gram filter (143.9 KiB
             * of table per workgroup) and 19.5 through the key image (76 KiB), although the filter scans a GiB in 0.28 ms against
             * 0.46.  The two cross near 30 MiB -- below SMH_ADAPT_MIN_BYTES, where launches do not report -- so a handle that keeps a
             * key image beside a big-table filter scans such texts with the image.  The handle's engine (A->engine) is left alone. */
            if (A && engine == SMH_ALGO_WM && wm->engine_forced < 0 && wm->keys && n < SMH_ADAPT_MIN_BYTES &&
                (wm->gram_kind == SMH_GRAM_BYTE_BIG || wm->gram_kind == SMH_GRAM_FLAT_BIG || wm->gram_kind == SMH_GRAM_FLAT4_BIG))
                return keys_launch(wm->keys, d_text, n, d_count, NULL, stream, adapt_arg(A, n, SMH_ENGINE_KEYS, stream));
            if (engine == SMH_ALGO_AC) return ac_launch_own(wm_automaton_engine(wm), d_text, n, d_count, stream, adapt_arg(A, n, engine, stream));
            if (engine == SMH_ENGINE_AC_FLAT) return ac_flat_launch(wm->flex_ac, d_text, n, d_count, stream, adapt_arg(A, n, engine, stream));
            if (engine == SMH_ENGINE_KEYS) return keys_launch(wm->keys, d_text, n, d_count, NULL, stream, adapt_arg(A, n, engine, stream));
            if (engine == SMH_ENGINE_HASH) return hash_launch(wm->hashes, d_text, n, d_count, NULL, stream, adapt_arg(A, n, engine, stream));
            return wm_launch_own(wm, d_text, n, d_count, stream, adapt_arg(wm_reports(wm) ? A : NULL, n, engine, stream), adapt_density(A, wm));
        }();
        const int rc_after = A ? adapt_order_after(A, stream) : SMH_OK;
        return rc != SMH_OK ? rc : rc_after;
    }
    if (variant != SMH_VARIANT_TABLE) {
        smh_set_error("smh_wm_scan: unknown variant %d", variant);
        return SMH_EINVAL;
    }
    smh_wm_dev *dv = NULL;
    int rc = wm_ensure_device(wm, &dv);
    if (rc != SMH_OK) return rc;
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    if ((rc = wm_ensure_reference_tables(wm, dv)) != SMH_OK) return rc;
    smh_wm_table_launch L;
    L.d_text = d_text; L.n = n; L.m = wm->m; L.shiftsize = wm->shiftsize; L.d_shift = dv->d_shift;
    L.d_bucket_off = dv->d_bucket_off; L.d_bucket = dv->d_bucket; L.d_pat_orig = dv->d_pat_orig;
    L.d_count = d_count; L.n_cus = n_cus;
    HIP_TRY(smh_launch_wm_table(L, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_wm_get_adapt(smh_wm *wm, smh_adapt_info *out)
{
    if (!wm || wm->magic != SMH_MAGIC_WM || !out || out->struct_size != sizeof *out) {
        smh_set_error("smh_wm_get_adapt: bad arguments (set struct_size = sizeof(smh_adapt_info))");
        return SMH_EINVAL;
    }
    double est[SMH_ENGINES];
    wm_estimates(wm, est);
    const int adaptive = wm_engines(wm) && wm->engine_forced < 0 && adapt_enabled();
    smh_adapt_dev *A = adapt_find(&wm->adapt);
    std::unique_lock<std::mutex> lock;
    if (A) lock = std::unique_lock<std::mutex>(*A->mu);
    if (A) adapt_poll(A);
    adapt_report(A, adaptive, wm_engine_static(wm), est, out);
    return SMH_OK;
}

static int wm_positions_impl(smh_wm *wm, int engine, const unsigned char *d_text, uint64_t n, uint64_t *d_positions, uint64_t capacity,
                             uint64_t *d_cursor, void *stream);
extern "C" int smh_wm_positions(smh_wm *wm, const unsigned char *d_text, uint64_t n, uint64_t *d_positions,
                                uint64_t capacity, uint64_t *d_cursor, void *stream)
{
    if (!wm || wm->magic != SMH_MAGIC_WM || !d_cursor || (capacity && !d_positions) || (n && !d_text)) {
        smh_set_error("smh_wm_positions: bad arguments");
        return SMH_EINVAL;
    }
    if (n < (uint64_t)wm->m) return SMH_OK;
    /* a handle that smh_wm_scan serialises and orders (same test as there) gets the same for its positions launches: they share the
     * per-device survivor queue with the scans and must not run beside them (round 6) */
    if (!(adapt_enabled() && (wm_engines(wm) || wm_reports(wm))))
        return wm_positions_impl(wm, wm_engine_static(wm), d_text, n, d_positions, capacity, d_cursor, stream);
    smh_adapt_dev *A = NULL;
    int rc = adapt_get(&wm->adapt, &A);
    if (rc != SMH_OK) return rc;
    const int engine = wm_engine_now(wm); /* (takes A->mu itself) */
    std::lock_guard<std::mutex> adapt_lock(*A->mu);
    if ((rc = adapt_order_before(A, stream)) != SMH_OK) return rc;
    rc = wm_positions_impl(wm, engine, d_text, n, d_positions, capacity, d_cursor, stream);
    const int rc_after = adapt_order_after(A, stream);
    return rc != SMH_OK ? rc : rc_after;
}
static int wm_positions_impl(smh_wm *wm, int engine, const unsigned char *d_text, uint64_t n, uint64_t *d_positions, uint64_t capacity,
                             uint64_t *d_cursor, void *stream)
{
    if (engine == SMH_ALGO_AC)
        return smh_ac_positions(wm_automaton_engine(wm), d_text, n, d_positions, capacity, d_cursor, stream);
    else if (engine == SMH_ENGINE_AC_FLAT && wm->flex_ac && wm->flex_ac->flat_ac)
        return ac_flat_positions(wm->flex_ac, d_text, n, d_positions, capacity, d_cursor, stream);
    else if (engine == SMH_ENGINE_KEYS && wm->keys && ((uintptr_t)d_text & 15u) == 0) {
        const smh_pos_out po = {d_positions, capacity, d_cursor};
        return keys_launch(wm->keys, d_text, n, NULL, &po, stream, smh_stats_arg{});
    } else if (engine == SMH_ENGINE_HASH && wm->hashes && ((uintptr_t)d_text & 15u) == 0) {
        const smh_pos_out po = {d_positions, capacity, d_cursor};
        return hash_launch(wm->hashes, d_text, n, NULL, &po, stream, smh_stats_arg{});
    }
    smh_wm_dev *dv = NULL;
    int rc = wm_ensure_device(wm, &dv);
    if (rc != SMH_OK) return rc;
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    if (((uintptr_t)d_text & 15u) == 0) {
        /* the tuned scan kernels in positions mode: matching columns are recorded as bits (exact
         * filters) or verified through the survivor queue, and appended per wave */
        smh_wm_launch L = {};
        L.d_text = d_text; L.n = n; L.m = wm->m; L.bits = wm->bits_per_symbol; L.block_symbols = wm->block_symbols;
        L.filter_log2 = wm->filter_log2; L.filter_hashed = wm->filter_hashed; L.filter_k = wm->filter_k; L.filter_le4 = wm->filter_le4; L.filter_exact = wm->filter_exact;
        L.d_filter = dv->d_filter; L.d_pair = dv->d_pair; L.verify_log2 = wm->verify_log2; L.d_verify = dv->d_verify; L.d_verify_ck = dv->d_verify_ck; L.ck_buckets = wm->ck_buckets; L.ck_seed = wm->ck_seed;
        L.d_gram = dv->d_gram; L.gram_kind = wm->gram_kind; L.gram_density = (float)wm->gram_density; L.gram_lane0 = (float)wm->gram_lane0; L.gram_planes = wm->gram_planes;
        if (wm->gram_kind == SMH_GRAM_FLAT || wm->gram_kind == SMH_GRAM_FLAT_BIG) L.gram_jb = wm->gram_jb;
        L.d_pat_sorted = dv->d_pat_sorted; L.d_queue = dv->d_queue; L.d_count = NULL; L.n_cus = n_cus;
        L.po.out = d_positions; L.po.capacity = capacity; L.po.cursor = d_cursor;
        HIP_TRY(smh_launch_wm_block_positions(L, (hipStream_t)stream));
        return SMH_OK;
    }
    /* unaligned text: the reference tables walked as given, one lane per 256 columns */
    if ((rc = wm_ensure_reference_tables(wm, dv)) != SMH_OK) return rc;
    smh_wm_table_launch L;
    L.d_text = d_text; L.n = n; L.m = wm->m; L.shiftsize = wm->shiftsize; L.d_shift = dv->d_shift;
    L.d_bucket_off = dv->d_bucket_off; L.d_bucket = dv->d_bucket; L.d_pat_orig = dv->d_pat_orig;
    L.d_count = NULL; L.n_cus = n_cus;
    HIP_TRY(smh_launch_wm_positions(L, d_positions, capacity, d_cursor, (hipStream_t)stream));
    return SMH_OK;
}

/* ---- mixed-length sets in one pass (pset_host.c): `suffix` is a handle compiled over the patterns'
 * last min-length symbols -- its block filter proposes END columns -- and `classes` are the per-length
 * handles whose verify tables decide them.  d_count / positions exactly as in the single-length calls. */
static int wm_multi_launch(smh_wm *suffix, smh_wm *const *classes, int n_classes, const unsigned char *d_text, uint64_t n,
                           uint64_t *d_count, const smh_pos_out *po, void *stream)
{
    if (!suffix || suffix->magic != SMH_MAGIC_WM || !classes || n_classes < 1 || n_classes > SMH_WM_MAX_CLASSES ||
        (n && !d_text)) {
        smh_set_error("smh_wm_scan_multi: bad arguments");
        return SMH_EINVAL;
    }
    if (((uintptr_t)d_text & 15u) != 0) {
        smh_set_error("smh_wm_scan_multi: d_text must be 16-byte aligned");
        return SMH_EINVAL;
    }
    if (n < (uint64_t)suffix->m) return SMH_OK;
    smh_wm_dev *sdv = NULL;
    int rc = wm_ensure_device(suffix, &sdv);
    if (rc != SMH_OK) return rc;
    smh_wm_class host[SMH_WM_MAX_CLASSES];
    for (int c = 0; c < n_classes; ++c) {
        smh_wm *k = classes[c];
        if (!k || k->magic != SMH_MAGIC_WM || k->m < suffix->m) {
            smh_set_error("smh_wm_scan_multi: class %d is not a Wu-Manber handle of length >= %d", c, suffix->m);
            return SMH_EINVAL;
        }
        smh_wm_dev *kdv = NULL;
        if ((rc = wm_ensure_device(k, &kdv)) != SMH_OK) return rc;
        if (!kdv->d_verify) {
            /* an exact class has no verify table of its own: build it now (host table exists only when the
             * filter is not exact) -- pset_host.c compiles the classes so that this cannot happen */
            smh_set_error("smh_wm_scan_multi: class %d has no verify table", c);
            return SMH_EUNSUP;
        }
        host[c].m = k->m;
        host[c].verify_log2 = k->verify_log2;
        host[c].verify = kdv->d_verify;
        host[c].pat_sorted = kdv->d_pat_sorted;
    }
    {
        /* The class table goes up when it has changed -- once per set and device (round 5; it used to go up with EVERY scan, from
         * this function's stack: 29 us in front of a 240 us kernel).  Round 6: with hipMemcpyAsync on the CALLER'S stream
         * out of the handle's own persistent copy (h_classes) and under a mutex of its own -- round 5 used a blocking copy on the
         * null stream with the process-wide g_dev_mu held, which (a) was an illegal synchronous call inside a stream capture,
         * (b) did not order behind launches on non-blocking streams and (c) stalled every other handle's ensure_device.  A pset's
         * scans must not overlap (include/smatcher_hip.h), so the stream that carries the scan is the stream the table must be
         * ordered on.  Inside a capture the copy becomes a node of the graph (it re-reads h_classes at every replay: same bytes)
         * and the table does not count as uploaded -- the first scan outside a capture still sends it. */
        static std::mutex classes_mu;
        std::lock_guard<std::mutex> lock(classes_mu);
        if (sdv->n_classes_up != n_classes || memcmp(sdv->h_classes, host, sizeof(smh_wm_class) * (size_t)n_classes) != 0) {
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
            memcpy(sdv->h_classes, host, sizeof(smh_wm_class) * (size_t)n_classes);
            HIP_TRY(hipMemcpyAsync(sdv->d_classes, sdv->h_classes, sizeof(smh_wm_class) * (size_t)n_classes, hipMemcpyHostToDevice, (hipStream_t)stream));
            sdv->n_classes_up = cap == hipStreamCaptureStatusNone ? n_classes : 0;
        }
    }
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    smh_wm *wm = suffix;
    smh_wm_launch L = {};
    L.d_text = d_text; L.n = n; L.m = wm->m; L.bits = wm->bits_per_symbol; L.block_symbols = wm->block_symbols;
    L.filter_log2 = wm->filter_log2; L.filter_hashed = wm->filter_hashed; L.filter_k = wm->filter_k; L.filter_le4 = wm->filter_le4;
    L.filter_exact = 0;
    L.d_filter = sdv->d_filter; L.d_pair = NULL; L.verify_log2 = 4; L.d_verify = NULL; L.d_pat_sorted = NULL;
    L.d_queue = sdv->d_queue; L.d_count = d_count; L.n_cus = n_cus;
    L.n_classes = n_classes; L.d_classes = sdv->d_classes;
    if (wm->gram_kind == SMH_GRAM_PAIR2) { /* grouped pair-gram filter over the full patterns (wm_host.c) */
        L.d_gram = sdv->d_gram; L.gram_kind = wm->gram_kind; L.gram_jb = wm->gram_jb;
        L.sfx_slot_off = wm->sfx_slot_off; L.sfx_ent_off = wm->sfx_ent_off; L.sfx_pat_off = wm->sfx_pat_off;
    }
    if (po) {
        L.po = *po;
        HIP_TRY(smh_launch_wm_block_positions(L, (hipStream_t)stream));
    } else {
        HIP_TRY(smh_launch_wm_block(L, (hipStream_t)stream));
    }
    return SMH_OK;
}

extern "C" int smh_wm_scan_multi(smh_wm *suffix, smh_wm *const *classes, int n_classes, const unsigned char *d_text,
                                 uint64_t n, uint64_t *d_count, void *stream)
{
    if (!d_count) { smh_set_error("smh_wm_scan_multi: bad arguments"); return SMH_EINVAL; }
    return wm_multi_launch(suffix, classes, n_classes, d_text, n, d_count, NULL, stream);
}

extern "C" int smh_wm_positions_multi(smh_wm *suffix, smh_wm *const *classes, int n_classes, const unsigned char *d_text,
                                      uint64_t n, uint64_t *d_positions, uint64_t capacity, uint64_t *d_cursor, void *stream)
{
    if (!d_cursor || (capacity && !d_positions)) { smh_set_error("smh_wm_positions_multi: bad arguments"); return SMH_EINVAL; }
    const smh_pos_out po = {d_positions, capacity, d_cursor};
    return wm_multi_launch(suffix, classes, n_classes, d_text, n, NULL, &po, stream);
}

extern "C" int smh_wm_count_host(smh_wm *wm, const unsigned char *text, uint64_t n, int variant, uint64_t *count,
                                 double *kernel_seconds)
{
    if (!wm || wm->magic != SMH_MAGIC_WM) { smh_set_error("smh_wm_count_host: bad handle"); return SMH_EINVAL; }
    return count_host(text, n, wm->m, count, kernel_seconds,
                      [&]() { return n < (uint64_t)wm->m ? SMH_OK : wm_prepare(wm, variant); },
                      [&](unsigned char *d_text, uint64_t len, uint64_t *d_count, void *st) { return smh_wm_scan(wm, d_text, len, d_count, variant, st); });
}

/* ------------------------------------------------------------------ mixed-length automaton (acm_host.c) */
struct smh_acm_dev {
    int device;
    smh_acm_dev *next;
    void *d_scan;
    uint32_t *d_goto;
    uint8_t *d_final;
    uint64_t *d_queue;
};

static void acm_dev_free_one(smh_acm_dev *dev)
{
    (void)hipFree(dev->d_scan);
    (void)hipFree(dev->d_goto);
    (void)hipFree(dev->d_final);
    (void)hipFree(dev->d_queue);
    delete dev;
}

extern "C" void smh_acm_dev_free(struct smh_acm_dev *dev) /* the whole list */
{
    while (dev) {
        smh_acm_dev *next = dev->next;
        acm_dev_free_one(dev);
        dev = next;
    }
}

extern "C" int smh_acm_scan(struct smh_acm *a, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream)
{
    if (!a || a->magic != SMH_MAGIC_ACM || !d_count || (n && !d_text) || ((uintptr_t)d_text & 15u) != 0) {
        smh_set_error("smh_acm_scan: bad arguments (the text must be 16-byte aligned)");
        return SMH_EINVAL;
    }
    if (n == 0) return SMH_OK;
    int n_cus = 0, rc;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    smh_acm_dev *dv = NULL;
    rc = ensure_device_set<smh_acm_dev>(&a->dev, acm_dev_free_one, [&](smh_acm_dev *d) -> int {
        int rc;
        if ((rc = upload(&d->d_scan, a->scan, (size_t)a->scan_bytes, 256 * 4)) != SMH_OK) return rc;
        if ((rc = upload((void **)&d->d_goto, a->g_goto, (size_t)a->nodes * (size_t)a->alphabet * 4, 256 * 4)) != SMH_OK) return rc;
        if ((rc = upload((void **)&d->d_final, a->g_final, (size_t)a->nodes, 16)) != SMH_OK) return rc;
        HIP_TRY(hipMalloc((void **)&d->d_queue, (size_t)smh_acm_max_blocks(n_cus) * (SMH_BLOCK_THREADS / 64) * SMH_ACM_QCAP * 8));
        return SMH_OK;
    }, &dv);
    if (rc != SMH_OK) return rc;
    smh_acm_launch L = {};
    L.C.text = d_text; L.C.n = n; L.C.K = a->K; L.C.max_len = a->max_len; L.C.sigma = a->alphabet;
    L.C.g_goto = dv->d_goto; L.C.g_final = dv->d_final;
    L.entry_bytes = a->entry_bytes; L.d_scan = dv->d_scan; L.lds_bytes = a->scan_bytes;
    L.d_queue = dv->d_queue; L.d_count = d_count; L.n_cus = n_cus;
    HIP_TRY(smh_launch_acm(L, (hipStream_t)stream));
    return SMH_OK;
}

/* ------------------------------------------------------------------ key engine (key_host.c, key_kernels.hip) */
struct smh_keys_dev {
    int device;
    smh_keys_dev *next;
    void *d_image;
};

static void keys_dev_free_one(smh_keys_dev *dev)
{
    (void)hipFree(dev->d_image);
    delete dev;
}

extern "C" void smh_keys_dev_free(struct smh_keys_dev *dev) /* the whole list */
{
    while (dev) {
        smh_keys_dev *next = dev->next;
        keys_dev_free_one(dev);
        dev = next;
    }
}

static int keys_ensure_device(struct smh_keys *k, smh_keys_dev **out)
{
    return ensure_device_set<smh_keys_dev>(&k->dev, keys_dev_free_one, [&](smh_keys_dev *d) -> int {
        return upload(&d->d_image, k->image, (size_t)k->P.bytes, 0);
    }, out);
}

static int keys_wg_per_cu()
{
    return smh_tune_int(SMH_TUNE_KEY, "wg=", 0);
}

/* one launch over [d_text, d_text + n): END columns counted into *d_count, or appended to po */
static int keys_launch(struct smh_keys *k, const unsigned char *d_text, uint64_t n, uint64_t *d_count, const smh_pos_out *po,
                       void *stream, const smh_stats_arg &SA)
{
    if (n < (uint64_t)k->m) return SMH_OK;
    int n_cus = 0, rc;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    smh_keys_dev *dv = NULL;
    if ((rc = keys_ensure_device(k, &dv)) != SMH_OK) return rc;
    smh_key_launch L = {};
    L.d_text = d_text; L.n = n; L.K = k->P; L.d_image = reinterpret_cast<const uint32_t *>(dv->d_image);
    L.d_count = d_count; L.n_cus = n_cus; L.wg_per_cu = keys_wg_per_cu(); L.stats = SA;
    if (smh_tune_has(SMH_TUNE_KEY, "noover=1")) L.K.bk_sentinel = 0xFFFFFFFEu; /* testing library only: no lane ever sees a sentinel -- the bucket image's scan without its overflow path, counts wrong (timing experiment) */
    if (po) {
        L.po = *po;
        HIP_TRY(smh_launch_keys_positions(L, (hipStream_t)stream));
    } else {
        HIP_TRY(smh_launch_keys(L, (hipStream_t)stream));
    }
    return SMH_OK;
}

extern "C" int smh_keys_scan(struct smh_keys *k, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream)
{
    if (!k || k->magic != SMH_MAGIC_KEYS || !d_count || (n && !d_text) || ((uintptr_t)d_text & 15u) != 0) {
        smh_set_error("smh_keys_scan: bad arguments (the text must be 16-byte aligned)");
        return SMH_EINVAL;
    }
    return keys_launch(k, d_text, n, d_count, NULL, stream, smh_stats_arg{});
}

extern "C" int smh_keys_positions(struct smh_keys *k, const unsigned char *d_text, uint64_t n, uint64_t *d_positions, uint64_t capacity,
                                  uint64_t *d_cursor, void *stream)
{
    if (!k || k->magic != SMH_MAGIC_KEYS || !d_cursor || (capacity && !d_positions) || (n && !d_text) || ((uintptr_t)d_text & 15u) != 0) {
        smh_set_error("smh_keys_positions: bad arguments (the text must be 16-byte aligned)");
        return SMH_EINVAL;
    }
    const smh_pos_out po = {d_positions, capacity, d_cursor};
    return keys_launch(k, d_text, n, NULL, &po, stream, smh_stats_arg{});
}

extern "C" int smh_keys_get_info(const struct smh_keys *k, smh_keys_info *out)
{
    const uint32_t r5_size = (uint32_t)offsetof(smh_keys_info, layout); /* the struct of round 5 ends in front of `layout` */
    if (!k || k->magic != SMH_MAGIC_KEYS || !out || (out->struct_size != sizeof *out && out->struct_size != r5_size)) {
        smh_set_error("smh_keys_get_info: bad arguments (set struct_size = sizeof(smh_keys_info))");
        return SMH_EINVAL;
    }
    out->alphabet = (uint32_t)k->alphabet; out->m = (uint32_t)k->m; out->keys = k->n_keys;
    out->key_bits = (uint32_t)(k->P.m * k->P.bits); out->slot_bytes = k->P.wide == 1 ? 8u : 4u;
    out->slots = k->P.slots; /* bucket image: buckets of the primary table (two slots each) */
    out->lds_bytes = k->P.bytes; out->est_ms_per_gib = k->ms_est;
    if (out->struct_size == sizeof *out) { out->layout = k->P.layout; out->overflow_keys = k->P.layout == 1 ? k->P.bk_overflow : 0u; }
    return SMH_OK;
}

extern "C" struct smh_keys *smh_keys_compile_patterns(const unsigned char *pattern_flat, int m, int p_size, int alphabet)
{
    const char *why = "";
    struct smh_keys *k = smh_keys_build(pattern_flat, m, p_size, alphabet, SMH_KEYS_LDS_BUDGET, &why);
    if (!k) smh_set_error("smh_keys_compile_patterns: the key engine does not take this set (%s)", why);
    return k;
}

/* ------------------------------------------------------------------ window-hash engine (hash_host.c, hash_kernels.hip) */
struct smh_hash_dev {
    int device;
    smh_hash_dev *next;
    void *d_bloom;
    void *d_table;
};

static void hash_dev_free_one(smh_hash_dev *dev)
{
    (void)hipFree(dev->d_bloom);
    (void)hipFree(dev->d_table);
    delete dev;
}

extern "C" void smh_hash_dev_free(struct smh_hash_dev *dev) /* the whole list */
{
    while (dev) {
        smh_hash_dev *next = dev->next;
        hash_dev_free_one(dev);
        dev = next;
    }
}

static int hash_ensure_device(struct smh_hashes *k, smh_hash_dev **out)
{
    return ensure_device_set<smh_hash_dev>(&k->dev, hash_dev_free_one, [&](smh_hash_dev *d) -> int {
        int rc = upload(&d->d_bloom, k->bloom, (size_t)k->P.bloom_bytes, 0);
        if (rc != SMH_OK) return rc;
        return upload(&d->d_table, k->table, (size_t)k->table_bytes, 64);
    }, out);
}

static int hash_launch(struct smh_hashes *k, const unsigned char *d_text, uint64_t n, uint64_t *d_count, const smh_pos_out *po,
                       void *stream, const smh_stats_arg &SA)
{
    if (n < (uint64_t)k->m) return SMH_OK;
    int n_cus = 0, rc;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    smh_hash_dev *dv = NULL;
    if ((rc = hash_ensure_device(k, &dv)) != SMH_OK) return rc;
    smh_hash_launch L = {};
    L.C.text = d_text; L.C.n = n; L.C.P = k->P; L.C.table = reinterpret_cast<const uint8_t *>(dv->d_table);
    L.C.drop = smh_tune_has(SMH_TUNE_HASH, "drop=1") ? 1u : 0u; /* testing library only: stage 1 alone, counts wrong (smh_tune.h) */
    L.d_bloom = reinterpret_cast<const uint32_t *>(dv->d_bloom); L.d_count = d_count; L.n_cus = n_cus; L.stats = SA;
    if (po) {
        L.po = *po;
        HIP_TRY(smh_launch_hash_positions(L, (hipStream_t)stream));
    } else {
        HIP_TRY(smh_launch_hash(L, (hipStream_t)stream));
    }
    return SMH_OK;
}

/* ------------------------------------------------------------------ SH */
struct smh_sh_dev {
    int device;
    smh_sh_dev *next;
    int32_t *d_transition;
    uint32_t *d_final;
    int32_t *d_bmbc; /* the table of the last scan (alphabet entries) */
};

static void sh_dev_free_one(smh_sh_dev *dev)
{
    (void)hipFree(dev->d_transition);
    (void)hipFree(dev->d_final);
    (void)hipFree(dev->d_bmbc);
    delete dev;
}

extern "C" void smh_sh_dev_free(struct smh_sh_dev *dev) /* the whole list */
{
    while (dev) {
        smh_sh_dev *next = dev->next;
        sh_dev_free_one(dev);
        dev = next;
    }
}

static int sh_ensure_device(struct smh_sh *sh, smh_sh_dev **out)
{
    return ensure_device_set<smh_sh_dev>(&sh->dev, sh_dev_free_one, [&](smh_sh_dev *d) -> int {
        int rc;
        const size_t A = (size_t)sh->alphabet;
        if ((rc = upload((void **)&d->d_transition, sh->g_transition, (size_t)sh->states * A * 4, 0)) != SMH_OK) return rc;
        if ((rc = upload((void **)&d->d_final, sh->g_final, (size_t)sh->states * 4, 0)) != SMH_OK) return rc;
        HIP_TRY(hipMalloc((void **)&d->d_bmbc, 256 * sizeof(int32_t)));
        return SMH_OK;
    }, out);
}

static int sh_prepare(struct smh_sh *sh, int variant)
{
    if (variant == SMH_VARIANT_TUNED) return sh->wm ? wm_prepare(sh->wm, variant) : ac_prepare(sh->ac, variant);
    smh_sh_dev *d = NULL;
    return sh_ensure_device(sh, &d);
}

extern "C" int smh_sh_scan(smh_sh *sh, const unsigned char *d_text, uint64_t n, const int *bmBc, uint64_t *d_count,
                           int variant, void *stream)
{
    if (!sh || sh->magic != SMH_MAGIC_SH || !d_count || (n && !d_text)) {
        smh_set_error("smh_sh_scan: bad arguments");
        return SMH_EINVAL;
    }
    int rc = smh_sh_check_bmbc(sh, bmBc);
    if (rc != SMH_OK) return rc;
    if (n < (uint64_t)sh->m) return SMH_OK;
    if (variant == SMH_VARIANT_TUNED) {
        /* every column is tested: the shifts of a valid table only skip columns that end no match */
        return sh->wm ? smh_wm_scan(sh->wm, d_text, n, d_count, SMH_VARIANT_TUNED, stream)
                      : smh_ac_scan(sh->ac, d_text, n, d_count, SMH_VARIANT_TUNED, stream);
    }
    if (variant != SMH_VARIANT_TABLE) {
        smh_set_error("smh_sh_scan: unknown variant %d", variant);
        return SMH_EINVAL;
    }
    smh_sh_dev *dv = NULL;
    if ((rc = sh_ensure_device(sh, &dv)) != SMH_OK) return rc;
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    /* the table travels with the launch (stream-ordered copy from a pageable host buffer: the call
     * returns after the copy has been staged) */
    HIP_TRY(hipMemcpyAsync(dv->d_bmbc, bmBc ? bmBc : sh->valid_bmbc, (size_t)sh->alphabet * sizeof(int32_t),
                           hipMemcpyHostToDevice, (hipStream_t)stream));
    smh_sh_table_launch L;
    L.d_text = d_text; L.n = n; L.m = sh->m; L.alphabet = sh->alphabet;
    L.d_transition = dv->d_transition; L.d_final = dv->d_final; L.d_bmbc = dv->d_bmbc;
    L.d_count = d_count; L.n_cus = n_cus;
    HIP_TRY(smh_launch_sh_table(L, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_sh_count_host(smh_sh *sh, const unsigned char *text, uint64_t n, const int *bmBc, int variant,
                                 uint64_t *count, double *kernel_seconds)
{
    if (!sh || sh->magic != SMH_MAGIC_SH) { smh_set_error("smh_sh_count_host: bad handle"); return SMH_EINVAL; }
    return count_host(text, n, sh->m, count, kernel_seconds,
                      [&]() { return n < (uint64_t)sh->m ? SMH_OK : sh_prepare(sh, variant); },
                      [&](unsigned char *d_text, uint64_t len, uint64_t *d_count, void *st) { return smh_sh_scan(sh, d_text, len, bmBc, d_count, variant, st); });
}

/* ------------------------------------------------------------------ SBOM */
struct smh_sbom_dev {
    int device;
    smh_sbom_dev *next;
    int32_t *d_transition;
    uint32_t *d_final_off;
    uint32_t *d_final_ids;
    uint8_t *d_patterns;
};

static void sbom_dev_free_one(smh_sbom_dev *dev)
{
    (void)hipFree(dev->d_transition);
    (void)hipFree(dev->d_final_off);
    (void)hipFree(dev->d_final_ids);
    (void)hipFree(dev->d_patterns);
    delete dev;
}

extern "C" void smh_sbom_dev_free(struct smh_sbom_dev *dev) /* the whole list */
{
    while (dev) {
        smh_sbom_dev *next = dev->next;
        sbom_dev_free_one(dev);
        dev = next;
    }
}

static int sbom_ensure_device(struct smh_sbom *sb, smh_sbom_dev **out)
{
    return ensure_device_set<smh_sbom_dev>(&sb->dev, sbom_dev_free_one, [&](smh_sbom_dev *d) -> int {
        int rc;
        const size_t A = (size_t)sb->alphabet;
        if ((rc = upload((void **)&d->d_transition, sb->g_transition, (size_t)sb->states * A * 4, 0)) != SMH_OK) return rc;
        if ((rc = upload((void **)&d->d_final_off, sb->g_final_off, ((size_t)sb->states + 1) * 4, 0)) != SMH_OK) return rc;
        if ((rc = upload((void **)&d->d_final_ids, sb->g_final_ids, (size_t)sb->listed * 4, 0)) != SMH_OK) return rc;
        return upload((void **)&d->d_patterns, sb->patterns, (size_t)sb->n_patterns * sb->m, 0);
    }, out);
}

static int sbom_prepare(struct smh_sbom *sb, int variant)
{
    if (variant == SMH_VARIANT_TUNED) return sb->wm ? wm_prepare(sb->wm, variant) : ac_prepare(sb->ac, variant);
    smh_sbom_dev *d = NULL;
    return sbom_ensure_device(sb, &d);
}

extern "C" int smh_sbom_scan(smh_sbom *sb, const unsigned char *d_text, uint64_t n, uint64_t *d_count, int variant,
                             void *stream)
{
    if (!sb || sb->magic != SMH_MAGIC_SBOM || !d_count || (n && !d_text)) {
        smh_set_error("smh_sbom_scan: bad arguments");
        return SMH_EINVAL;
    }
    if (n < (uint64_t)sb->m) return SMH_OK;
    if (variant == SMH_VARIANT_TUNED)
        return sb->wm ? smh_wm_scan(sb->wm, d_text, n, d_count, SMH_VARIANT_TUNED, stream)
                      : smh_ac_scan(sb->ac, d_text, n, d_count, SMH_VARIANT_TUNED, stream);
    if (variant != SMH_VARIANT_TABLE) {
        smh_set_error("smh_sbom_scan: unknown variant %d", variant);
        return SMH_EINVAL;
    }
    smh_sbom_dev *dv = NULL;
    int rc = sbom_ensure_device(sb, &dv);
    if (rc != SMH_OK) return rc;
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    smh_sbom_table_launch L;
    L.d_text = d_text; L.n = n; L.m = sb->m; L.alphabet = sb->alphabet;
    L.d_transition = dv->d_transition; L.d_final_off = dv->d_final_off; L.d_final_ids = dv->d_final_ids;
    L.d_patterns = dv->d_patterns; L.d_count = d_count; L.n_cus = n_cus;
    HIP_TRY(smh_launch_sbom_table(L, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_sbom_count_host(smh_sbom *sb, const unsigned char *text, uint64_t n, int variant, uint64_t *count,
                                   double *kernel_seconds)
{
    if (!sb || sb->magic != SMH_MAGIC_SBOM) { smh_set_error("smh_sbom_count_host: bad handle"); return SMH_EINVAL; }
    return count_host(text, n, sb->m, count, kernel_seconds,
                      [&]() { return n < (uint64_t)sb->m ? SMH_OK : sbom_prepare(sb, variant); },
                      [&](unsigned char *d_text, uint64_t len, uint64_t *d_count, void *st) { return smh_sbom_scan(sb, d_text, len, d_count, variant, st); });
}

/* ------------------------------------------------------------------ SOG */
struct smh_sog_dev {
    int device;
    smh_sog_dev *next;
    uint8_t *d_t8;
    uint32_t *d_hs;
    int32_t *d_index;
    uint8_t *d_hs2;
    uint8_t *d_patterns;
};

static void sog_dev_free_one(smh_sog_dev *dev)
{
    (void)hipFree(dev->d_t8);
    (void)hipFree(dev->d_hs);
    (void)hipFree(dev->d_index);
    (void)hipFree(dev->d_hs2);
    (void)hipFree(dev->d_patterns);
    delete dev;
}

extern "C" void smh_sog_dev_free(struct smh_sog_dev *dev) /* the whole list */
{
    while (dev) {
        smh_sog_dev *next = dev->next;
        sog_dev_free_one(dev);
        dev = next;
    }
}

static int sog_ensure_device(struct smh_sog *sg, smh_sog_dev **out)
{
    return ensure_device_set<smh_sog_dev>(&sg->dev, sog_dev_free_one, [&](smh_sog_dev *d) -> int {
        int rc;
        if ((rc = upload((void **)&d->d_t8, sg->t8, (size_t)1 << 24, 0)) != SMH_OK) return rc;
        if ((rc = upload((void **)&d->d_hs, sg->hs, (size_t)sg->n_patterns * 4, 0)) != SMH_OK) return rc;
        if ((rc = upload((void **)&d->d_index, sg->index, (size_t)sg->n_patterns * 4, 0)) != SMH_OK) return rc;
        if ((rc = upload((void **)&d->d_hs2, sg->hs2, 8192, 0)) != SMH_OK) return rc;
        return upload((void **)&d->d_patterns, sg->patterns, (size_t)sg->n_patterns * 8, 0);
    }, out);
}

static int sog_prepare(struct smh_sog *sg, int variant)
{
    if (variant == SMH_VARIANT_TUNED) return wm_prepare(sg->wm, variant);
    smh_sog_dev *d = NULL;
    return sog_ensure_device(sg, &d);
}

extern "C" int smh_sog_scan(smh_sog *sg, const unsigned char *d_text, uint64_t n, uint64_t *d_count, int variant, void *stream)
{
    if (!sg || sg->magic != SMH_MAGIC_SOG || !d_count || (n && !d_text)) {
        smh_set_error("smh_sog_scan: bad arguments");
        return SMH_EINVAL;
    }
    if (n < 8) return SMH_OK;
    if (variant == SMH_VARIANT_TUNED) return smh_wm_scan(sg->wm, d_text, n, d_count, SMH_VARIANT_TUNED, stream);
    if (variant != SMH_VARIANT_TABLE) {
        smh_set_error("smh_sog_scan: unknown variant %d", variant);
        return SMH_EINVAL;
    }
    smh_sog_dev *dv = NULL;
    int rc = sog_ensure_device(sg, &dv);
    if (rc != SMH_OK) return rc;
    int n_cus = 0;
    if ((rc = current_cus(&n_cus)) != SMH_OK) return rc;
    smh_sog_table_launch L;
    L.d_text = d_text; L.n = n; L.d_t8 = dv->d_t8; L.d_hs = dv->d_hs; L.d_index = dv->d_index; L.d_hs2 = dv->d_hs2;
    L.d_patterns = dv->d_patterns; L.p_size = (int)sg->n_patterns; L.d_count = d_count; L.n_cus = n_cus;
    HIP_TRY(smh_launch_sog_table(L, (hipStream_t)stream));
    return SMH_OK;
}

extern "C" int smh_sog_count_host(smh_sog *sg, const unsigned char *text, uint64_t n, int variant, uint64_t *count,
                                  double *kernel_seconds)
{
    if (!sg || sg->magic != SMH_MAGIC_SOG) { smh_set_error("smh_sog_count_host: bad handle"); return SMH_EINVAL; }
    return count_host(text, n, 8, count, kernel_seconds,
                      [&]() { return n < 8 ? SMH_OK : sog_prepare(sg, variant); },
                      [&](unsigned char *d_text, uint64_t len, uint64_t *d_count, void *st) { return smh_sog_scan(sg, d_text, len, d_count, variant, st); });
}

/* ------------------------------------------------------------------ legacy names (smatcher.h) */
static void die_with_error(const char *where)
{
    fprintf(stderr, "%s: %s\n", where, smh_last_error());
    exit(1);
}

/* smatcher.h:90 / ac/ac.c:198-222 -- same count, computed by ac_dfa_kernel */
extern "C" unsigned search_ac(unsigned char *text, int n, struct ac_table *table)
{
    struct smh_ac_table_box *box = (struct smh_ac_table_box *)table;
    if (!box || box->magic != SMH_MAGIC_AC) fail("search_ac: not a table from preproc_ac\n");
    uint64_t count = 0;
    if (smh_ac_count_host(box->ac, text, n < 0 ? 0 : (uint64_t)n, SMH_VARIANT_TUNED, &count, NULL) != SMH_OK)
        die_with_error("search_ac");
    return (unsigned)count;
}

/* The legacy GPU names take the caller's TABLES with every call (cuda/cuda_ac.cu:594, cuda/cuda_wm.cu:183) and main.c:583-592,
 * 623-648 calls the five variants back to back on the same tables.  Until round 5 every call compiled a handle from them, sent
 * its table set to the device and freed it.  Round 6: the last handle of each family is kept, keyed on the caller's pointers and
 * shapes AND a 128-bit digest of everything the reference's search reads from them (the transition / supply / final arrays; the patterns,
 * SHIFT, PREFIX_size and the used entries of PREFIX_value / PREFIX_index) -- a caller that rewrites its arrays in place gets a new
 * handle.  smh_host_path_release() frees the kept handles; smh_legacy_handle_builds() counts the compiles (tests, bench). */
struct legacy_digest { /* 128 bits: two independent multiply-xorshift lanes over the same words */
    uint64_t a, b;
    bool operator==(const legacy_digest &o) const { return a == o.a && b == o.b; }
};
static legacy_digest digest_words(legacy_digest h, const void *data, size_t bytes)
{
    const unsigned char *p = (const unsigned char *)data;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        memcpy(&w, p + i, 8);
        h.a = (h.a ^ w) * 0x9E3779B97F4A7C15ull;
        h.a ^= h.a >> 29;
        h.b = (h.b + w) * 0xD6E8FEB86659FD93ull;
        h.b ^= h.b >> 32;
    }
    uint64_t tail = 0;
    if (i < bytes) memcpy(&tail, p + i, bytes - i);
    h.a = (h.a ^ tail ^ ((uint64_t)bytes << 56)) * 0xBF58476D1CE4E5B9ull;
    h.a ^= h.a >> 32;
    h.b = (h.b + tail + (uint64_t)bytes) * 0x94D049BB133111EBull;
    h.b ^= h.b >> 31;
    return h;
}
struct legacy_key {
    const void *ptr[5];
    int m, p_size, alphabet;
    legacy_digest digest;
    bool operator==(const legacy_key &o) const { return memcmp(ptr, o.ptr, sizeof ptr) == 0 && m == o.m && p_size == o.p_size && alphabet == o.alphabet && digest == o.digest; }
};
static std::mutex g_legacy_mu; /* the legacy names are the reference's single-threaded API: one call at a time */
static smh_ac *g_legacy_ac = NULL;
static smh_wm *g_legacy_wm = NULL;
static legacy_key g_legacy_ac_key, g_legacy_wm_key;
static std::atomic<uint64_t> g_legacy_builds{0};
extern "C" uint64_t smh_legacy_handle_builds(void) { return g_legacy_builds.load(); }
static void legacy_release(void)
{
    std::lock_guard<std::mutex> lock(g_legacy_mu);
    if (g_legacy_ac) smh_ac_free(g_legacy_ac);
    if (g_legacy_wm) smh_wm_free(g_legacy_wm);
    g_legacy_ac = NULL;
    g_legacy_wm = NULL;
}

static void cuda_ac_any(int k, int variant, int m, unsigned char *text, int n, int p_size, int alphabet,
                        int *state_transition, unsigned int *state_supply, unsigned int *state_final)
{
    std::lock_guard<std::mutex> lock(g_legacy_mu);
    const uint64_t rows = (uint64_t)m * (uint64_t)p_size + 1u;
    legacy_key key = {{state_transition, state_supply, state_final, NULL, NULL}, m, p_size, alphabet, {0, 0}};
    if (state_transition && state_supply && state_final && m > 0 && p_size > 0 && alphabet > 0) {
        key.digest = digest_words(legacy_digest{0x5EED, 0xFACADE}, state_transition, (size_t)rows * (size_t)alphabet * sizeof(int));
        key.digest = digest_words(key.digest, state_supply, (size_t)rows * sizeof(unsigned int));
        key.digest = digest_words(key.digest, state_final, (size_t)rows * sizeof(unsigned int));
    }
    if (!g_legacy_ac || !(key == g_legacy_ac_key)) {
        if (g_legacy_ac) smh_ac_free(g_legacy_ac);
        g_legacy_ac = smh_ac_compile_tables(state_transition, state_supply, state_final, rows, alphabet, m);
        g_legacy_ac_key = key;
        g_legacy_builds++;
    }
    smh_ac *ac = g_legacy_ac;
    if (!ac) die_with_error("cuda_ac");
    uint64_t count = 0;
    double secs = 0.0;
    if (smh_ac_count_host(ac, text, n < 0 ? 0 : (uint64_t)n, variant, &count, &secs) != SMH_OK)
        die_with_error("cuda_ac");
    /* cuda/cuda_ac.cu:675 */
    printf("Kernel %d matches \t%i\t time \t%f\n", k, (int)count, secs);
    fflush(stdout);
}

#define SMH_CUDA_AC(K, VARIANT)                                                                            \
    extern "C" void cuda_ac##K(int m, unsigned char *text, int n, int p_size, int alphabet,                \
                               int *state_transition, unsigned int *state_supply, unsigned int *state_final) \
    {                                                                                                      \
        cuda_ac_any(K, VARIANT, m, text, n, p_size, alphabet, state_transition, state_supply, state_final); \
    }
SMH_CUDA_AC(1, SMH_VARIANT_TABLE)
SMH_CUDA_AC(2, SMH_VARIANT_TABLE)
SMH_CUDA_AC(3, SMH_VARIANT_TUNED)
SMH_CUDA_AC(4, SMH_VARIANT_TUNED)
SMH_CUDA_AC(5, SMH_VARIANT_TUNED)

/* the caller's alphabet is not an argument of search_wu*: recover it from the global
 * shiftsize = (alphabet-1)*21+1 that wu_determine_shiftsize set (wu/wu.c:18-47) */
static int alphabet_from_shiftsize(void)
{
    static const int known[] = {2, 4, 8, 20, 128, 256, 512, 1024};
    for (size_t i = 0; i < sizeof known / sizeof known[0]; ++i)
        if (smh_wu_shiftsize_for(known[i]) == shiftsize) return known[i];
    fail("search_wu: call wu_determine_shiftsize first\n");
    return 0;
}

static unsigned int wm_any(const unsigned char *flat, int m, int p_size, int alphabet, unsigned char *text, int n,
                           int *SHIFT, int *PREFIX_value, int *PREFIX_index, int *PREFIX_size, int variant,
                           double *secs, const void *pattern_identity)
{
    std::lock_guard<std::mutex> lock(g_legacy_mu);
    legacy_key key = {{pattern_identity, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size}, m, p_size, alphabet, {0, 0}};
    const uint32_t buckets = alphabet > 0 ? smh_wu_shiftsize_for(alphabet) : 0u;
    if (flat && SHIFT && PREFIX_value && PREFIX_index && PREFIX_size && m > 0 && p_size > 0 && buckets) {
        key.digest = digest_words(legacy_digest{0x5EED, 0xFACADE}, flat, (size_t)m * (size_t)p_size);
        key.digest = digest_words(key.digest, SHIFT, (size_t)buckets * sizeof(int));
        key.digest = digest_words(key.digest, PREFIX_size, (size_t)buckets * sizeof(int));
        for (uint32_t h = 0; h < buckets; ++h) { /* the entries search_wu reads: PREFIX_size[h] of them per bucket (wu/wu.c:76-80) */
            const int used = PREFIX_size[h] < 0 ? 0 : (PREFIX_size[h] > p_size ? p_size : PREFIX_size[h]);
            if (!used) continue;
            key.digest = digest_words(key.digest, PREFIX_value + (size_t)h * (size_t)p_size, (size_t)used * sizeof(int));
            key.digest = digest_words(key.digest, PREFIX_index + (size_t)h * (size_t)p_size, (size_t)used * sizeof(int));
        }
    }
    if (!g_legacy_wm || !(key == g_legacy_wm_key)) {
        if (g_legacy_wm) smh_wm_free(g_legacy_wm);
        g_legacy_wm = smh_wm_compile_tables(flat, m, p_size, alphabet, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size);
        g_legacy_wm_key = key;
        g_legacy_builds++;
    }
    smh_wm *wm = g_legacy_wm;
    if (!wm) die_with_error("wu-manber");
    uint64_t count = 0;
    if (smh_wm_count_host(wm, text, n < 0 ? 0 : (uint64_t)n, variant, &count, secs) != SMH_OK)
        die_with_error("wu-manber");
    return (unsigned int)count;
}

/* smatcher.h:106 / wu/wu.c:151-209 */
extern "C" unsigned int search_wu2(unsigned char *pattern_flat, int m, int p_size, unsigned char *text, int n,
                                   int *SHIFT, int *PREFIX_value, int *PREFIX_index, int *PREFIX_size)
{
    return wm_any(pattern_flat, m, p_size, alphabet_from_shiftsize(), text, n, SHIFT, PREFIX_value, PREFIX_index,
                  PREFIX_size, SMH_VARIANT_TUNED, NULL, pattern_flat);
}

/* smatcher.h:105 / wu/wu.c:49-107 */
extern "C" unsigned int search_wu(unsigned char **pattern, int m, int p_size, unsigned char *text, int n,
                                  int *SHIFT, int *PREFIX_value, int *PREFIX_index, int *PREFIX_size)
{
    std::vector<unsigned char> flat((size_t)m * (size_t)p_size);
    for (int j = 0; j < p_size; ++j) memcpy(flat.data() + (size_t)j * m, pattern[j], (size_t)m);
    return wm_any(flat.data(), m, p_size, alphabet_from_shiftsize(), text, n, SHIFT, PREFIX_value, PREFIX_index,
                  PREFIX_size, SMH_VARIANT_TUNED, NULL, pattern);
}

#define SMH_CUDA_WM(K, VARIANT)                                                                              \
    extern "C" int cuda_wm##K(unsigned char *pattern_flat, int m, unsigned char *text, int n, int p_size,    \
                              int alphabet, int B, int *SHIFT, int *PREFIX_value, int *PREFIX_index,         \
                              int *PREFIX_size, double *gpuTime)                                             \
    {                                                                                                        \
        (void)B;                                                                                             \
        double secs = 0.0;                                                                                   \
        unsigned int c = wm_any(pattern_flat, m, p_size, alphabet, text, n, SHIFT, PREFIX_value, PREFIX_index, \
                                PREFIX_size, VARIANT, &secs, pattern_flat);                                  \
        if (gpuTime) *gpuTime = secs; /* cuda/cuda_wm.cu:302 */                                              \
        return (int)c;                                                                                       \
    }
SMH_CUDA_WM(1, SMH_VARIANT_TABLE)
SMH_CUDA_WM(2, SMH_VARIANT_TABLE)
SMH_CUDA_WM(3, SMH_VARIANT_TUNED)
SMH_CUDA_WM(4, SMH_VARIANT_TUNED)
SMH_CUDA_WM(5, SMH_VARIANT_TUNED)

/* smatcher.h:94 / sh/sh.c:151-176 -- same count, computed by the tuned kernels */
extern "C" unsigned search_sh(int m, unsigned char *text, int n, struct ac_table *table, int *bmBc)
{
    struct smh_sh_table_box *box = (struct smh_sh_table_box *)table;
    if (!box || box->magic != SMH_MAGIC_SH) fail("search_sh: not a table from preproc_sh\n");
    if (m != box->sh->m) fail("search_sh: m differs from the m given to preproc_sh\n");
    uint64_t count = 0;
    if (smh_sh_count_host(box->sh, text, n < 0 ? 0 : (uint64_t)n, bmBc, SMH_VARIANT_TUNED, &count, NULL) != SMH_OK)
        die_with_error("search_sh");
    return (unsigned)count;
}

static void cuda_sh_any(int k, int variant, int m, unsigned char *text, int n, int p_size, int alphabet,
                        int *state_transition, unsigned int *state_final, int *bmBc)
{
    smh_sh *sh = smh_sh_compile_tables(state_transition, state_final, (uint64_t)m * p_size + 1, alphabet, m);
    if (!sh) die_with_error("cuda_sh");
    uint64_t count = 0;
    double secs = 0.0;
    if (smh_sh_count_host(sh, text, n < 0 ? 0 : (uint64_t)n, bmBc, variant, &count, &secs) != SMH_OK)
        die_with_error("cuda_sh");
    /* cuda/cuda_sh.cu:191 */
    printf("Kernel %d matches \t%i\t time \t%f\n", k, (int)count, secs);
    fflush(stdout);
    smh_sh_free(sh);
}

#define SMH_CUDA_SH(K, VARIANT)                                                                        \
    extern "C" void cuda_sh##K(int m, unsigned char *text, int n, int p_size, int alphabet,            \
                               int *state_transition, unsigned int *state_final, int *bmBc)            \
    {                                                                                                  \
        cuda_sh_any(K, VARIANT, m, text, n, p_size, alphabet, state_transition, state_final, bmBc);    \
    }
SMH_CUDA_SH(1, SMH_VARIANT_TABLE)
SMH_CUDA_SH(2, SMH_VARIANT_TABLE)
SMH_CUDA_SH(3, SMH_VARIANT_TUNED)
SMH_CUDA_SH(4, SMH_VARIANT_TUNED)
SMH_CUDA_SH(5, SMH_VARIANT_TUNED)

/* smatcher.h:98 / sbom/sbom.c:128-172 -- same count, computed by the tuned kernels.  `pattern` is the
 * array preproc_sbom was given (the reference compares against it; the handle holds its own copy). */
extern "C" unsigned search_sbom(unsigned char **pattern, int m, unsigned char *text, int n, struct sbom_table *table)
{
    (void)pattern;
    struct smh_sbom_table_box *box = (struct smh_sbom_table_box *)table;
    if (!box || box->magic != SMH_MAGIC_SBOM) fail("search_sbom: not a table from preproc_sbom\n");
    if (m != box->sb->m) fail("search_sbom: m differs from the m given to preproc_sbom\n");
    uint64_t count = 0;
    if (smh_sbom_count_host(box->sb, text, n < 0 ? 0 : (uint64_t)n, SMH_VARIANT_TUNED, &count, NULL) != SMH_OK)
        die_with_error("search_sbom");
    return (unsigned)count;
}

static void cuda_sbom_any(int k, int variant, unsigned char *pattern_flat, int m, unsigned char *text, int n, int p_size,
                          int alphabet, int *state_transition, unsigned int *state_final_multi)
{
    smh_sbom *sb = smh_sbom_compile_tables(pattern_flat, m, p_size, alphabet, state_transition, state_final_multi,
                                           (uint64_t)m * p_size + 1);
    if (!sb) die_with_error("cuda_sbom");
    uint64_t count = 0;
    double secs = 0.0;
    if (smh_sbom_count_host(sb, text, n < 0 ? 0 : (uint64_t)n, variant, &count, &secs) != SMH_OK)
        die_with_error("cuda_sbom");
    /* cuda/cuda_sbom.cu:212 */
    printf("Kernel %d matches \t%i\t time \t%f\n", k, (int)count, secs);
    fflush(stdout);
    smh_sbom_free(sb);
}

#define SMH_CUDA_SBOM(K, VARIANT)                                                                              \
    extern "C" void cuda_sbom##K(unsigned char *pattern, int m, unsigned char *text, int n, int p_size,        \
                                 int alphabet, int *state_transition, unsigned int *state_final_multi)         \
    {                                                                                                          \
        cuda_sbom_any(K, VARIANT, pattern, m, text, n, p_size, alphabet, state_transition, state_final_multi); \
    }
SMH_CUDA_SBOM(1, SMH_VARIANT_TABLE)
SMH_CUDA_SBOM(2, SMH_VARIANT_TABLE)
SMH_CUDA_SBOM(3, SMH_VARIANT_TUNED)
SMH_CUDA_SBOM(4, SMH_VARIANT_TUNED)
SMH_CUDA_SBOM(5, SMH_VARIANT_TUNED)

/* smatcher.h:109 / sog/sog8.c:97-115 -- the number of 8-byte windows that equal a pattern, computed by the tuned kernels */
static uint64_t sog_any(int variant, uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2,
                        const unsigned char *pattern_flat, int m, unsigned char *text, int n, int p_size, double *secs)
{
    if (m != 8) fail("SOG is built for patterns of length 8 (sog/sog8.c)\n");
    smh_sog *sg = smh_sog_compile_tables(T8, scanner_hs, scanner_index, scanner_hs2, pattern_flat, p_size);
    if (!sg) die_with_error("sog");
    uint64_t count = 0;
    if (smh_sog_count_host(sg, text, n < 0 ? 0 : (uint64_t)n, variant, &count, secs) != SMH_OK) die_with_error("sog");
    smh_sog_free(sg);
    return count;
}

extern "C" unsigned int search_sog8(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2,
                                    unsigned char **pattern, int m, unsigned char *text, int n, int p_size, int B)
{
    (void)B;
    if (m != 8 || p_size < 1 || !pattern) fail("search_sog8: bad arguments (m must be 8)\n");
    std::vector<unsigned char> flat((size_t)p_size * 8);
    for (int j = 0; j < p_size; ++j) memcpy(flat.data() + (size_t)j * 8, pattern[j], 8);
    return (unsigned int)sog_any(SMH_VARIANT_TUNED, T8, scanner_hs, scanner_index, scanner_hs2, flat.data(), m, text, n, p_size, NULL);
}

#define SMH_CUDA_SOG(K, VARIANT)                                                                                      \
    extern "C" void cuda_sog##K(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2,          \
                                unsigned char *pattern, int m, unsigned char *text, int n, int p_size, int B)         \
    {                                                                                                                  \
        (void)B;                                                                                                       \
        double secs = 0.0;                                                                                             \
        const uint64_t c = sog_any(VARIANT, T8, scanner_hs, scanner_index, scanner_hs2, pattern, m, text, n, p_size, &secs); \
        printf("Kernel %d matches \t%i\t time \t%f\n", K, (int)c, secs); /* cuda/cuda_sog.cu:314 */                  \
        fflush(stdout);                                                                                                \
    }
SMH_CUDA_SOG(1, SMH_VARIANT_TABLE)
SMH_CUDA_SOG(2, SMH_VARIANT_TABLE)
SMH_CUDA_SOG(3, SMH_VARIANT_TUNED)
SMH_CUDA_SOG(4, SMH_VARIANT_TUNED)
SMH_CUDA_SOG(5, SMH_VARIANT_TUNED)
