/*
 * csrc/smh_multi.hip -- one process, all GPUs of a node: the reference driver's MPI layer for this path,
 * natively.
 *
 * The reference spreads a text over ranks with MPI_Scatterv, every rank scans its byte range plus an m-1
 * halo, and one MPI_Reduce(MPI_INT, MPI_SUM) adds the counts (main.c:464-489, 654-657).  Here ONE host
 * process owns every device: per device a stream, a resident text shard (the main.c:467-477 ranges with the
 * true length of the last one) and a 64-bit counter; a scan call launches the tuned kernel on every device
 * (asynchronously, so the devices run side by side) and the counts meet in ONE ncclAllReduce(ncclUint64,
 * ncclSum) over an RCCL communicator created with ncclCommInitAll -- 8 bytes over xGMI, the only exchange.
 * Handles keep one table set per device (smh_runtime.hip), so the same compiled automaton serves all shards.
 *
 * RCCL is bound at run time (dlopen of librccl.so.1: the library a PyTorch process has already loaded, or
 * ROCm's), so that single-GPU users of libsmatcher_hip.so do not load it.  Without it smh_multi_create fails
 * unless the caller allows the host-side sum (flag SMH_MULTI_HOST_SUM), which is what the CPU-only build
 * check uses.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "smh_internal.h"

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            smh_set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SMH_ENODEV;                                                            \
        }                                                                                 \
    } while (0)

/* the five RCCL entry points this file uses */
struct smh_rccl_api {
    void *lib;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    const char *(*GetErrorString)(ncclResult_t);
};
static smh_rccl_api g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load(void)
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (g_rccl.lib) return SMH_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *lib = NULL;
    for (size_t i = 0; i < sizeof names / sizeof names[0] && !lib; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        smh_set_error("smh_multi: RCCL not found (%s)", dlerror());
        return SMH_ENODEV;
    }
    smh_rccl_api a = {};
    a.lib = lib;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(lib, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.GroupStart = (decltype(a.GroupStart))dlsym(lib, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(lib, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.AllReduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString) {
        smh_set_error("smh_multi: librccl lacks an expected symbol");
        dlclose(lib);
        return SMH_ENODEV;
    }
    g_rccl = a;
    return SMH_OK;
}

#define NCCL_TRY(expr)                                                                        \
    do {                                                                                      \
        ncclResult_t r_ = (expr);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            smh_set_error("%s: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            return SMH_ENODEV;                                                                \
        }                                                                                     \
    } while (0)

struct smh_multi_dev {
    int device;
    hipStream_t stream;
    unsigned char *d_text; /* this device's byte range + halo, 16-byte aligned, 64 readable bytes of slack */
    uint64_t begin;        /* first byte of the range in the whole text */
    uint64_t bytes;        /* bytes resident: range + min(halo, what follows it) */
    uint64_t *d_count;     /* [0] this device's count, [1] the all-reduced sum */
    ncclComm_t comm;
};

/* a handle as it was when it was warmed up: its address can be reused after a free and its plan or engine can be changed
 * (smh_ac_set_scan_plan, smh_*_set_scan_engine), its serial cannot and every such change bumps the generation */
struct smh_warm_key {
    uint64_t serial;
    uint32_t generation;
    bool operator==(const smh_warm_key &o) const { return serial == o.serial && generation == o.generation; }
};
extern "C" void smh_dev_set_slot(int slot); /* smh_runtime.hip */

struct smh_multi {
    uint32_t magic;
    int n;
    int rccl;
    uint64_t n_total; /* length of the whole text */
    uint64_t per;     /* ceil(n_total / n): main.c:375-378 */
    int halo;         /* bytes kept beyond every range: scans with m - 1 <= halo are possible */
    std::vector<smh_multi_dev> dev;
    int share;        /* SMH_MULTI_SHARE_DEVICE: logical shards may sit on one card (per-shard table sets by slot: smh_dev_set_slot) */
    std::vector<smh_warm_key> warmed; /* handles (serial, plan generation) whose kernels have run once on every device (prepare_all) */
};
#define SMH_MAGIC_MULTI 0x4d554c54u /* "MULT" */

static int check(const smh_multi *mg, const char *who)
{
    if (!mg || mg->magic != SMH_MAGIC_MULTI) {
        smh_set_error("%s: bad handle", who);
        return SMH_EINVAL;
    }
    return SMH_OK;
}

extern "C" void smh_multi_free(smh_multi *mg)
{
    if (!mg || mg->magic != SMH_MAGIC_MULTI) return;
    for (auto &d : mg->dev) {
        if (hipSetDevice(d.device) != hipSuccess) continue;
        if (d.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d.comm);
        (void)hipFree(d.d_text);
        (void)hipFree(d.d_count);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
    mg->magic = 0;
    delete mg;
}

extern "C" int smh_multi_create(smh_multi **out, const int *devices, int n_devices, int flags)
{
    if (!out || n_devices < 1 || n_devices > SMH_MULTI_MAX_DEVICES) {
        smh_set_error("smh_multi_create: bad arguments");
        return SMH_EINVAL;
    }
    *out = NULL;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess) visible = 0;
    /* the one-card rehearsal of the N-device flow: logical shard i on device i mod (visible devices), every shard with a
     * stream, a text range, a counter and -- through the runtime's slot keys -- table sets of its own; counts are added on
     * the host (a communicator takes a device once).  The flag, or SMH_MULTI_SHARE_DEVICE=1 in the environment. */
    if (const char *e = getenv("SMH_MULTI_SHARE_DEVICE"); e && atoi(e) != 0) flags |= SMH_MULTI_SHARE_DEVICE;
    const int share = (flags & SMH_MULTI_SHARE_DEVICE) != 0;
    std::vector<int> ids(n_devices);
    bool twice = false;
    for (int i = 0; i < n_devices; ++i) {
        ids[i] = devices ? devices[i] : (share && visible > 0 ? i % visible : i);
        if (ids[i] < 0 || ids[i] >= visible) {
            smh_set_error("smh_multi_create: device %d of %d asked for, %d visible", ids[i], n_devices, visible);
            return SMH_ENODEV;
        }
        for (int j = 0; j < i; ++j)
            if (ids[j] == ids[i]) {
                if (!share) {
                    smh_set_error("smh_multi_create: device %d listed twice (an RCCL communicator takes a device once; "
                                  "SMH_MULTI_SHARE_DEVICE rehearses several shards on one card)", ids[i]);
                    return SMH_EINVAL;
                }
                twice = true;
            }
    }
    int rccl = 1;
    if (twice || (flags & SMH_MULTI_NO_RCCL)) rccl = 0;
    if (rccl && rccl_load() != SMH_OK) {
        if (!(flags & SMH_MULTI_HOST_SUM)) return SMH_ENODEV;
        rccl = 0;
    }
    smh_multi *mg = new smh_multi();
    mg->magic = SMH_MAGIC_MULTI;
    mg->n = n_devices;
    mg->rccl = rccl;
    mg->share = share;
    mg->n_total = 0;
    mg->per = 0;
    mg->halo = 0;
    mg->dev.resize(n_devices);
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int i = 0; i < n_devices; ++i) {
        smh_multi_dev &d = mg->dev[i];
        d = smh_multi_dev{};
        d.device = ids[i];
        hipError_t e = hipSetDevice(d.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&d.d_count, 64);
        if (e == hipSuccess) e = hipMemset(d.d_count, 0, 64);
        if (e != hipSuccess) {
            smh_set_error("smh_multi_create: device %d: %s", d.device, hipGetErrorString(e));
            smh_multi_free(mg);
            (void)hipSetDevice(prev);
            return SMH_ENODEV;
        }
    }
    if (rccl) {
        std::vector<ncclComm_t> comms(n_devices);
        const ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_devices, ids.data());
        if (r != ncclSuccess) {
            smh_set_error("ncclCommInitAll(%d devices): %s", n_devices, g_rccl.GetErrorString(r));
            smh_multi_free(mg);
            (void)hipSetDevice(prev);
            return SMH_ENODEV;
        }
        for (int i = 0; i < n_devices; ++i) mg->dev[i].comm = comms[i];
    }
    (void)hipSetDevice(prev);
    *out = mg;
    return SMH_OK;
}

extern "C" int smh_multi_device_count(const smh_multi *mg) { return mg && mg->magic == SMH_MAGIC_MULTI ? mg->n : 0; }
extern "C" int smh_multi_uses_rccl(const smh_multi *mg) { return mg && mg->magic == SMH_MAGIC_MULTI ? mg->rccl : 0; }

/* (re)allocate every device's shard for a text of n_total bytes; ranges as main.c:467-477 */
static int place(smh_multi *mg, uint64_t n_total, int halo)
{
    if (halo < 0) halo = 0;
    mg->n_total = n_total;
    mg->per = (n_total + (uint64_t)mg->n - 1) / (uint64_t)mg->n;
    mg->halo = halo;
    mg->warmed.clear();
    for (int i = 0; i < mg->n; ++i) {
        smh_multi_dev &d = mg->dev[i];
        HIP_TRY(hipSetDevice(d.device));
        (void)hipFree(d.d_text);
        d.d_text = NULL;
        d.begin = (uint64_t)i * mg->per;
        if (d.begin > n_total) d.begin = n_total;
        uint64_t end = d.begin + mg->per + (uint64_t)halo;
        if (end > n_total) end = n_total;
        d.bytes = end - d.begin;
        HIP_TRY(hipMalloc((void **)&d.d_text, ((d.bytes + 15) / 16) * 16 + 64));
    }
    return SMH_OK;
}

extern "C" int smh_multi_load_text(smh_multi *mg, const unsigned char *text, uint64_t n, int halo)
{
    int rc = check(mg, "smh_multi_load_text");
    if (rc != SMH_OK) return rc;
    if (n && !text) { smh_set_error("smh_multi_load_text: NULL text"); return SMH_EINVAL; }
    int prev = 0;
    (void)hipGetDevice(&prev);
    rc = place(mg, n, halo);
    for (int i = 0; rc == SMH_OK && i < mg->n; ++i) {
        smh_multi_dev &d = mg->dev[i];
        if (hipSetDevice(d.device) != hipSuccess ||
            (d.bytes && hipMemcpyAsync(d.d_text, text + d.begin, d.bytes, hipMemcpyHostToDevice, d.stream) != hipSuccess)) {
            smh_set_error("smh_multi_load_text: copy to device %d failed", d.device);
            rc = SMH_ENODEV;
        }
    }
    for (int i = 0; i < mg->n; ++i)
        if (hipSetDevice(mg->dev[i].device) == hipSuccess) (void)hipStreamSynchronize(mg->dev[i].stream);
    (void)hipSetDevice(prev);
    return rc;
}

extern "C" int smh_multi_generate_text(smh_multi *mg, uint64_t n_total, uint64_t seed, int alphabet, int halo)
{
    int rc = check(mg, "smh_multi_generate_text");
    if (rc != SMH_OK) return rc;
    int prev = 0;
    (void)hipGetDevice(&prev);
    rc = place(mg, n_total, halo);
    for (int i = 0; rc == SMH_OK && i < mg->n; ++i) {
        smh_multi_dev &d = mg->dev[i];
        if (hipSetDevice(d.device) != hipSuccess) { smh_set_error("smh_multi_generate_text: hipSetDevice"); rc = SMH_ENODEV; break; }
        /* every shard is generated where it will be scanned: nothing crosses PCIe or xGMI */
        rc = smh_corpus_text_device(d.d_text, d.bytes, d.begin, seed, alphabet, d.stream);
    }
    for (int i = 0; i < mg->n; ++i)
        if (hipSetDevice(mg->dev[i].device) == hipSuccess) (void)hipStreamSynchronize(mg->dev[i].stream);
    (void)hipSetDevice(prev);
    return rc;
}

/* Everything the first scan of a handle would otherwise do inside the timed region, on every device AT ONCE: one
 * host thread per device builds the handle's table set there (smh_runtime.hip ensure_device_set: hipMalloc +
 * synchronous copies, outside the process-wide mutex) and runs one scan of the first few KiB of the shard into a
 * scratch counter, which loads the kernel's code object on that device and fills the per-device launch-attribute
 * cache.  After it *seconds of a count call is launches + reduce only (the reference times the kernel alone,
 * cuda/cuda_wm.cu:271-283).  The count calls run it themselves before their clock starts; calling it ahead of time
 * makes the first count call as fast as the tenth. */
template <typename Prep, typename Scan>
static int prepare_all(smh_multi *mg, const smh_warm_key &key, int m, Prep prep, Scan scan)
{
    if (m - 1 > mg->halo && mg->n_total) {
        smh_set_error("smh_multi: pattern length %d needs a halo of %d bytes, the text was placed with %d", m, m - 1, mg->halo);
        return SMH_EINVAL;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    bool warm = false; /* the table sets are looked up every time (a list walk); the warm-up scan runs once per handle */
    for (const smh_warm_key &k : mg->warmed) warm = warm || k == key;
    std::vector<int> rcs((size_t)mg->n, SMH_OK);
    std::vector<std::string> errs((size_t)mg->n);
    auto work = [&](int i) {
        smh_multi_dev &d = mg->dev[i];
        int rc = SMH_OK;
        if (hipSetDevice(d.device) != hipSuccess) {
            smh_set_error("smh_multi: hipSetDevice(%d) failed", d.device);
            rc = SMH_ENODEV;
        }
        smh_dev_set_slot(mg->share ? i + 1 : 0);
        if (rc == SMH_OK) rc = prep();
        if (rc == SMH_OK && !warm && d.d_text && d.bytes >= (uint64_t)m) {
            const uint64_t len = d.bytes < 16384u ? d.bytes : 16384u;
            rc = scan(d.d_text, len, d.d_count + 2, (void *)d.stream); /* slot 2: scratch, never read */
            if (rc == SMH_OK && hipStreamSynchronize(d.stream) != hipSuccess) {
                smh_set_error("smh_multi: device %d: warm-up scan failed: %s", d.device, hipGetErrorString(hipGetLastError()));
                rc = SMH_ENODEV;
            }
        }
        smh_dev_set_slot(0);
        rcs[(size_t)i] = rc;
        if (rc != SMH_OK) errs[(size_t)i] = smh_last_error(); /* the message is thread-local: carry it over */
    };
    if (mg->n == 1 || warm) { /* every device has its table set: a list walk each, not worth a thread */
        for (int i = 0; i < mg->n; ++i) work(i);
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < mg->n; ++i) th.emplace_back(work, i);
        for (auto &t : th) t.join();
    }
    (void)hipSetDevice(prev);
    for (int i = 0; i < mg->n; ++i)
        if (rcs[(size_t)i] != SMH_OK) {
            smh_set_error("%s", errs[(size_t)i].c_str());
            return rcs[(size_t)i];
        }
    if (!warm && mg->n_total) mg->warmed.push_back(key);
    return SMH_OK;
}

/* launch `scan(device text, shard length, device counter, stream)` on every device, reduce, read back */
template <typename Prep, typename Scan>
static int count_all(smh_multi *mg, const smh_warm_key &key, int m, uint64_t *total, uint64_t *per_device, double *seconds, Prep prep, Scan scan)
{
    if (!total) { smh_set_error("smh_multi: NULL result"); return SMH_EINVAL; }
    if (m - 1 > mg->halo) {
        smh_set_error("smh_multi: pattern length %d needs a halo of %d bytes, the text was placed with %d", m, m - 1, mg->halo);
        return SMH_EINVAL;
    }
    {
        /* table sets that are already there cost a list walk per device; missing ones go up now, before the clock */
        const int prc = prepare_all(mg, key, m, prep, scan);
        if (prc != SMH_OK) return prc;
    }
    int prev = 0;
    (void)hipGetDevice(&prev);
    int rc = SMH_OK;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; rc == SMH_OK && i < mg->n; ++i) {
        smh_multi_dev &d = mg->dev[i];
        /* shard i scans [begin, min(begin + per + m - 1, n)): the true length, where the reference passes the
         * padded one (main.c:376,630) */
        uint64_t len = mg->per + (uint64_t)(m - 1);
        if (d.begin + len > mg->n_total) len = mg->n_total - d.begin;
        if (hipSetDevice(d.device) != hipSuccess || hipMemsetAsync(d.d_count, 0, 16, d.stream) != hipSuccess) {
            smh_set_error("smh_multi: device %d: hipSetDevice / memset failed", d.device);
            rc = SMH_ENODEV;
            break;
        }
        smh_dev_set_slot(mg->share ? i + 1 : 0);
        rc = scan(d.d_text, len, d.d_count, (void *)d.stream); /* asynchronous: the devices scan side by side */
        smh_dev_set_slot(0);
    }
    if (rc == SMH_OK && mg->rccl) {
        /* the MPI_Reduce of main.c:656 as one RCCL all-reduce of a 64-bit count per device */
        ncclResult_t r = g_rccl.GroupStart();
        for (int i = 0; r == ncclSuccess && i < mg->n; ++i) {
            smh_multi_dev &d = mg->dev[i];
            r = g_rccl.AllReduce(d.d_count, d.d_count + 1, 1, ncclUint64, ncclSum, d.comm, d.stream);
        }
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = re;
        if (r != ncclSuccess) {
            smh_set_error("ncclAllReduce: %s", g_rccl.GetErrorString(r));
            rc = SMH_ENODEV;
        }
    }
    uint64_t host[SMH_MULTI_MAX_DEVICES][2];
    memset(host, 0, sizeof host);
    for (int i = 0; i < mg->n; ++i) {
        smh_multi_dev &d = mg->dev[i];
        if (hipSetDevice(d.device) != hipSuccess) { rc = rc == SMH_OK ? SMH_ENODEV : rc; continue; }
        if (hipMemcpyAsync(host[i], d.d_count, 16, hipMemcpyDeviceToHost, d.stream) != hipSuccess ||
            hipStreamSynchronize(d.stream) != hipSuccess) {
            if (rc == SMH_OK) { smh_set_error("smh_multi: device %d: %s", d.device, hipGetErrorString(hipGetLastError())); rc = SMH_ENODEV; }
        }
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    (void)hipSetDevice(prev);
    if (rc != SMH_OK) return rc;
    uint64_t sum = 0;
    for (int i = 0; i < mg->n; ++i) {
        sum += host[i][0];
        if (per_device) per_device[i] = host[i][0];
    }
    if (mg->rccl) {
        for (int i = 0; i < mg->n; ++i)
            if (host[i][1] != sum) { /* every rank holds the reduced total, and it is the sum of the shard counts */
                smh_set_error("smh_multi: device %d holds %llu after the all-reduce, the shard counts add up to %llu",
                              mg->dev[i].device, (unsigned long long)host[i][1], (unsigned long long)sum);
                return SMH_ENODEV;
            }
    }
    *total = sum;
    return SMH_OK;
}

extern "C" int smh_multi_ac_count(smh_multi *mg, smh_ac *ac, uint64_t *total, uint64_t *per_device, double *seconds)
{
    int rc = check(mg, "smh_multi_ac_count");
    if (rc != SMH_OK) return rc;
    smh_ac_info info;
    if ((rc = smh_ac_get_info(ac, &info)) != SMH_OK) return rc;
    return count_all(mg, smh_warm_key{ac->serial, ac->generation}, (int)info.m, total, per_device, seconds, [&]() { return smh_ac_prepare_device(ac); },
                     [&](unsigned char *t, uint64_t len, uint64_t *c, void *s) {
                         return smh_ac_scan(ac, t, len, c, SMH_VARIANT_TUNED, s);
                     });
}

extern "C" int smh_multi_wm_count(smh_multi *mg, smh_wm *wm, uint64_t *total, uint64_t *per_device, double *seconds)
{
    int rc = check(mg, "smh_multi_wm_count");
    if (rc != SMH_OK) return rc;
    smh_wm_info info;
    if ((rc = smh_wm_get_info(wm, &info)) != SMH_OK) return rc;
    return count_all(mg, smh_warm_key{wm->serial, wm->generation}, (int)info.m, total, per_device, seconds, [&]() { return smh_wm_prepare_device(wm); },
                     [&](unsigned char *t, uint64_t len, uint64_t *c, void *s) {
                         return smh_wm_scan(wm, t, len, c, SMH_VARIANT_TUNED, s);
                     });
}

extern "C" int smh_multi_ac_prepare(smh_multi *mg, smh_ac *ac)
{
    int rc = check(mg, "smh_multi_ac_prepare");
    if (rc != SMH_OK) return rc;
    smh_ac_info info;
    if ((rc = smh_ac_get_info(ac, &info)) != SMH_OK) return rc;
    return prepare_all(mg, smh_warm_key{ac->serial, ac->generation}, (int)info.m, [&]() { return smh_ac_prepare_device(ac); },
                       [&](unsigned char *t, uint64_t len, uint64_t *c, void *s) {
                           return smh_ac_scan(ac, t, len, c, SMH_VARIANT_TUNED, s);
                       });
}

extern "C" int smh_multi_wm_prepare(smh_multi *mg, smh_wm *wm)
{
    int rc = check(mg, "smh_multi_wm_prepare");
    if (rc != SMH_OK) return rc;
    smh_wm_info info;
    if ((rc = smh_wm_get_info(wm, &info)) != SMH_OK) return rc;
    return prepare_all(mg, smh_warm_key{wm->serial, wm->generation}, (int)info.m, [&]() { return smh_wm_prepare_device(wm); },
                       [&](unsigned char *t, uint64_t len, uint64_t *c, void *s) {
                           return smh_wm_scan(wm, t, len, c, SMH_VARIANT_TUNED, s);
                       });
}
