/*
 * csrc/smh_launch.h -- launch entry points of the kernel translation units,
 * called by smh_runtime.hip.  All take a hipStream_t and return hipError_t.
 */
#ifndef SMH_LAUNCH_H
#define SMH_LAUNCH_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include "ac_lane.h"
#include "wm_lane.h"
#include "acm_lane.h"
#include "corpus_gen.h"
#include "key_hash.h"
#include "hash_lane.h"
#include "smh_stats.h"
#include "smh_tune.h" /* development knobs: constants unless -DSMH_TESTING */

#define SMH_BLOCK_THREADS 1024
#define SMH_LDS_BUDGET (156u * 1024u) /* of the 160 KiB per CU; the rest is left to the runtime */
#define SMH_LDS_MIN 256u               /* smallest dynamic LDS a scan kernel is launched with: its final reduction uses the first 128 bytes */
#define SMH_MAX_HALO_CHUNKS 4          /* fast paths cover m - 1 <= 64 */
#define SMH_DEPTH_FIRST_MIN 72         /* depth_first[] is padded to max(this, m + 2): indexed with h + 1 <= 65 and t + 1 <= m */

/*
 * Launch attributes of one kernel instantiation, per device.
 * hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device's copy of the code
 * object, and it and the occupancy query cost tens of microseconds of host time during which the GPU
 * idles between the caller's events.  So: the attribute is raised ONCE per device to the whole 160 KiB
 * (every LDS size a plan can ask for is then launchable, and two handles with different table sizes do
 * not undo each other's setting), and the occupancy answer is remembered per (device, LDS size).
 * One instance per launch_one<> instantiation (a function-local static); guarded by a mutex because
 * the multi-device driver (smh_multi.hip) launches from one host thread per GPU.
 */
#define SMH_MAX_DEVICES 16
#define SMH_LDS_ATTR_MAX (160u * 1024u)
struct smh_attr_cache {
    struct slot { int device; uint32_t lds_bytes; int per_cu; };
    std::mutex mu;
    bool attr_set[SMH_MAX_DEVICES] = {};
    slot e[4 * SMH_MAX_DEVICES];
    int used = 0, victim = 0;
    /* per_cu = workgroups of `block_threads` threads that fit one CU with `lds_bytes` of dynamic LDS */
    template <typename K>
    hipError_t get(K kern, uint32_t lds_bytes, int block_threads, int *per_cu)
    {
        int dev = 0;
        hipError_t err = hipGetDevice(&dev);
        if (err != hipSuccess) return err;
        std::lock_guard<std::mutex> lock(mu);
        if (dev < 0 || dev >= SMH_MAX_DEVICES || !attr_set[dev]) {
            err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)SMH_LDS_ATTR_MAX);
            if (err != hipSuccess) return err;
            if (dev >= 0 && dev < SMH_MAX_DEVICES) attr_set[dev] = true;
        }
        for (int i = 0; i < used; ++i)
            if (e[i].device == dev && e[i].lds_bytes == lds_bytes) {
                *per_cu = e[i].per_cu;
                return hipSuccess;
            }
        int q = 0;
        err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, block_threads, lds_bytes);
        if (err != hipSuccess) return err;
        const int at = used < 4 * SMH_MAX_DEVICES ? used++ : (victim = (victim + 1) % (4 * SMH_MAX_DEVICES));
        e[at] = slot{dev, lds_bytes, q};
        *per_cu = q;
        return hipSuccess;
    }
};

struct smh_ac_launch {
    smh_ac_verify_ctx V;        /* text, n, m, K, sigma, full DFA, depth_first, trunc1 (device pointers) */
    smh_ac_df df;               /* depth_first[0..71] by value (kernel argument) */
    int stride;                 /* 1 or 2 */
    uint32_t full_rows;         /* stride 2: 0 = plain image, else the hybrid image's full-row count */
    int exact;                  /* K == m: flags are matches, no queue */
    int scan_entry_bytes;
    const void *d_scan_table;   /* LDS image */
    uint32_t lds_bytes;         /* multiple of 16 */
    uint64_t *d_queue;          /* smh_ac_max_blocks * 16 waves * SMH_AC_QCAP entries (NULL when exact) */
    uint64_t *d_count;
    int n_cus;
    uint64_t *d_wave_times;     /* development aid: 3 ticks per wave (ac_kernels.inc), else NULL */
    smh_stats_arg stats;        /* st != NULL: the launch reports its candidates and its duration (smh_stats.h) */
};
uint32_t smh_ac_max_blocks(int n_cus);
hipError_t smh_launch_ac_dfa(const smh_ac_launch &L, hipStream_t stream);
hipError_t smh_launch_ac_dfa_positions(const smh_ac_launch &L, hipStream_t stream);
hipError_t smh_launch_ac_dfa_wide(const smh_ac_launch &L, hipStream_t stream);
bool smh_lds_oob_probe(int n_cus);   /* ac_kernels.inc: the per-device probe behind the unclamped hybrid kernels; blocking, run where a table set is built */
bool smh_lds_oob_reads_zero(void);   /* the cached answer (false while the device has not been probed) */

struct smh_ac_table_launch {
    const uint8_t *d_text;
    uint64_t n;
    int m;
    int alphabet;
    const int32_t *d_transition;
    const uint32_t *d_supply;
    const uint32_t *d_final;
    uint64_t *d_count;
    int n_cus;
};
hipError_t smh_launch_ac_table(const smh_ac_table_launch &L, hipStream_t stream);
hipError_t smh_launch_ac_positions(const smh_ac_verify_ctx &V, uint64_t *d_positions, uint64_t capacity,
                                   uint64_t *d_cursor, int n_cus, hipStream_t stream);

struct smh_acm_launch {
    smh_acm_ctx C;          /* text, n, K, max_len, sigma, goto trie (device pointers) */
    int entry_bytes;
    const void *d_scan;     /* LDS image */
    uint32_t lds_bytes;     /* multiple of 16 */
    uint64_t *d_queue;      /* smh_acm_max_blocks * 16 waves * SMH_ACM_QCAP entries */
    uint64_t *d_count;
    int n_cus;
};
uint32_t smh_acm_max_blocks(int n_cus);
hipError_t smh_launch_acm(const smh_acm_launch &L, hipStream_t stream);

struct smh_sh_table_launch {
    const uint8_t *d_text;
    uint64_t n;
    int m;
    int alphabet;
    const int32_t *d_transition; /* reversed trie, reference layout */
    const uint32_t *d_final;
    const int32_t *d_bmbc;       /* alphabet entries */
    uint64_t *d_count;
    int n_cus;
};
hipError_t smh_launch_sh_table(const smh_sh_table_launch &L, hipStream_t stream);

struct smh_sbom_table_launch {
    const uint8_t *d_text;
    uint64_t n;
    int m;
    int alphabet;
    const int32_t *d_transition;  /* factor oracle, reference layout */
    const uint32_t *d_final_off;  /* [states + 1] */
    const uint32_t *d_final_ids;
    const uint8_t *d_patterns;    /* n_patterns * m */
    uint64_t *d_count;
    int n_cus;
};
hipError_t smh_launch_sbom_table(const smh_sbom_table_launch &L, hipStream_t stream);

struct smh_sog_table_launch {
    const uint8_t *d_text;
    uint64_t n;
    const uint8_t *d_t8;       /* 2^24 bytes */
    const uint32_t *d_hs;      /* sorted pattern hashes */
    const int32_t *d_index;
    const uint8_t *d_hs2;      /* 8192 bytes */
    const uint8_t *d_patterns; /* p_size * 8 */
    int p_size;
    uint64_t *d_count;
    int n_cus;
};
hipError_t smh_launch_sog_table(const smh_sog_table_launch &L, hipStream_t stream);

struct smh_wm_launch {
    const uint8_t *d_text;
    uint64_t n;
    int m;
    int bits;
    int block_symbols;
    int filter_log2;
    int filter_hashed;
    int filter_k;
    int filter_le4;
    int filter_exact;
    const uint32_t *d_filter;
    const uint32_t *d_pair; /* pair filter (alphabet 4, m <= 8, exact), else NULL */
    const uint32_t *d_gram; /* gram filter (128 KiB), else NULL; gram_kind 1 = symbol pairs (alphabet 4), 2 = byte grams */
    int gram_kind;
    float gram_density;     /* surviving columns per text column the compile measured on random text */
    float gram_lane0;       /* pair form: columns per wave-chunk let through by the assumption made for lane 0 alone */
    int gram_planes;        /* pair form: planes J (2..15) */
    int gram_jb;            /* gram_kind 4 (grouped pairs, mixed-length sets): planes of the short-pattern group */
    uint32_t sfx_slot_off, sfx_ent_off, sfx_pat_off; /* gram_kind 4: the verify stage's suffix index inside d_gram (byte offsets; 0 = none) */
    int verify_log2;
    const uint32_t *d_verify;
    const uint32_t *d_verify_ck; /* the cuckoo form of the verify entries (smh_internal.h), else NULL */
    uint32_t ck_buckets, ck_seed;
    const uint8_t *d_pat_sorted;
    uint64_t *d_queue; /* smh_wm_max_blocks * 16 waves * SMH_WM_QCAP columns (NULL when exact) */
    uint64_t *d_count;
    int n_cus;
    smh_pos_out po;       /* positions mode only */
    int n_classes;        /* > 0: mixed-length set in one pass, d_classes[n_classes] on the device */
    const smh_wm_class *d_classes;
    smh_stats_arg stats;  /* st != NULL: the launch reports its surviving columns and its duration (smh_stats.h) */
    int verify_mode;      /* gram kernels: 0 = choose from gram_density, 1 = staged verify, 2 = in-register verify (the runtime's
                           * choice from the survivors it has measured on this text) */
};
uint32_t smh_wm_max_blocks(int n_cus);
hipError_t smh_launch_wm_block(const smh_wm_launch &L, hipStream_t stream);
hipError_t smh_launch_wm_block_positions(const smh_wm_launch &L, hipStream_t stream);

struct smh_wm_table_launch {
    const uint8_t *d_text;
    uint64_t n;
    int m;
    uint32_t shiftsize;
    const uint16_t *d_shift;     /* SHIFT narrowed to 16 bits */
    const uint32_t *d_bucket_off;
    const int32_t *d_bucket;
    const uint8_t *d_pat_orig;
    uint64_t *d_count;
    int n_cus;
};
hipError_t smh_launch_wm_table(const smh_wm_table_launch &L, hipStream_t stream);
hipError_t smh_launch_wm_positions(const smh_wm_table_launch &L, uint64_t *d_positions, uint64_t capacity,
                                   uint64_t *d_cursor, hipStream_t stream);

struct smh_key_launch { /* key engine (key_kernels.hip) */
    const uint8_t *d_text;
    uint64_t n;
    smh_key_params K;
    const uint32_t *d_image; /* K.bytes */
    uint64_t *d_count;
    int n_cus;
    int wg_per_cu;        /* 0 = the default (one) */
    smh_pos_out po;       /* positions mode only */
    smh_stats_arg stats;  /* st != NULL: the launch reports its duration (smh_stats.h) */
};
hipError_t smh_launch_keys(const smh_key_launch &L, hipStream_t stream);
hipError_t smh_launch_keys_positions(const smh_key_launch &L, hipStream_t stream);

struct smh_hash_launch { /* window-hash engine (hash_kernels.hip) */
    smh_hash_ctx C;          /* text, n, parameters, pattern table (device pointers) */
    const uint32_t *d_bloom; /* C.P.bloom_bytes: the LDS image */
    uint64_t *d_count;
    int n_cus;
    smh_pos_out po;          /* positions mode only */
    smh_stats_arg stats;     /* st != NULL: the launch reports its surviving columns and its duration (smh_stats.h) */
};
hipError_t smh_launch_hash(const smh_hash_launch &L, hipStream_t stream);
hipError_t smh_launch_hash_positions(const smh_hash_launch &L, hipStream_t stream);

hipError_t smh_launch_corpus_text(uint8_t *d_out, uint64_t n, uint64_t offset, uint64_t seed,
                                  int alphabet, hipStream_t stream);
hipError_t smh_launch_corpus_text_kind(uint8_t *d_out, uint64_t n, uint64_t offset, uint64_t seed, int alphabet, int kind,
                                       const smh_corpus_tabs &T, hipStream_t stream);

#endif
