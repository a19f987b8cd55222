/*
 * include/smatcher_hip.h -- extended C ABI of libsmatcher_hip.so.
 *
 * smatcher.h keeps the reference's host API verbatim (int n, unsigned count,
 * one blocking call per scan that re-uploads the text, as cuda/cuda_ac.cu:627
 * and cuda/cuda_wm.cu:232 do).  This header is the documented superset the
 * BASELINE configurations need and the reference cannot express:
 *
 *   - 64-bit text lengths and match counts (reference: int n smatcher.h:90,105;
 *     unsigned / int results; MPI_INT reduce main.c:656);
 *   - text that is already resident in GPU memory (device pointer in, device
 *     counter out, caller's stream) so one upload serves AC and WM scans and
 *     the timed region contains the kernel only (what the reference times:
 *     cuda/cuda_ac.cu:642-663, cuda/cuda_wm.cu:263-286);
 *   - compiled, reusable automaton / table handles (the reference rebuilds
 *     device state in every cuda_* call, cuda/cuda_ac.cu:594-689);
 *   - the byte-range shard formula of main.c:375-378,464-477 with true shard
 *     lengths, for one-process-per-GPU drivers that sum counts with RCCL.
 *
 * Text and pattern bytes are symbol codes 0 .. alphabet-1, as in the reference (which indexes next[alphabet]
 * with the raw byte, ac/ac.c:136,209, and is undefined beyond it).  Here a text byte >= alphabet is never used as
 * an out-of-range index -- every table address is masked or clamped -- but WHICH windows containing such a
 * byte count as matches is unspecified and may differ between scan plans and engines; validate the text
 * where that matters.  Pattern symbols >= alphabet are refused at compile time (SMH_EINVAL).
 *
 * Plain C types only; `stream` arguments are hipStream_t passed as void*
 * (NULL = the default stream).  Functions returning int return SMH_OK (0) or a
 * negative SMH_E* code and leave a message for smh_last_error(); nothing here
 * exits the process (the legacy names in smatcher.h do, like the reference).
 */
#ifndef SMATCHER_HIP_H
#define SMATCHER_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMH_OK 0
#define SMH_EINVAL (-1)  /* bad argument (symbol >= alphabet, m < 1, unaligned text, ...) */
#define SMH_ENODEV (-2)  /* no usable HIP device / HIP runtime error */
#define SMH_ENOMEM (-3)
#define SMH_EUNSUP (-4)  /* outside the supported envelope */

/* kernel families selectable per scan */
#define SMH_VARIANT_TUNED 0 /* LDS-resident DFA (AC) / block-filter + verify (WM) */
#define SMH_VARIANT_TABLE 1 /* walks the reference-layout tables as given (goto/supply/final; SHIFT/PREFIX buckets) */

const char *smh_version(void);
const char *smh_last_error(void);

/* ---- runtime / device memory (so that C callers need no HIP headers) ---- */
int smh_device_count(void); /* 0 when there is no GPU or no HIP runtime */
int smh_set_device(int device);
int smh_device_name(char *buf, size_t cap);
/* "domain:bus:device.function" of the CURRENT device (hipDeviceGetPCIBusId): what tells two ranks' cards apart in a record */
int smh_device_pci_bus_id(char *buf, size_t cap);
int smh_device_malloc(void **dptr, uint64_t bytes);
int smh_device_free(void *dptr);
int smh_device_memset(void *dptr, int value, uint64_t bytes, void *stream);
int smh_copy_to_device(void *dst, const void *src, uint64_t bytes, void *stream);
int smh_copy_to_host(void *dst, const void *src, uint64_t bytes, void *stream);
int smh_stream_synchronize(void *stream);
/* measurement aid: a pure streaming read of d_buf[0, bytes) -- 16-byte loads, XOR of all
 * words into *d_out (device uint64) -- so that a bench can report what a read-only kernel reaches on
 * the same buffer in the same run.  d_buf must be 16-byte aligned. */
int smh_stream_read_probe(const void *d_buf, uint64_t bytes, uint64_t *d_out, void *stream);
/* other shapes of the same read (csrc/smh_runtime.hip, tools/readsweep.hip): variant 0 = the kernel above, 1 = 4 KiB
 * wave-chunks at 8 waves per CU, 2 = grid-stride at 8 waves per CU, 3 = two wave-chunks in flight at 4 waves per CU,
 * 4 = wave-chunks from the LDS counter at 16 waves per CU (how the scan kernels run).  A bench reports the best of
 * them as "what a streaming read reaches on this device". */
int smh_stream_read_probe_variant(const void *d_buf, uint64_t bytes, uint64_t *d_out, void *stream, int variant);

/* The blocking *_count_host helpers and the legacy names of smatcher.h (search_ac, search_wu, cuda_* ...) take the text
 * as a pageable host buffer.  It crosses PCIe in 64 MiB pieces through two device buffers of a pooled, grow-only workspace
 * (piece k+1 copies while piece k is scanned; pieces overlap by m - 1 bytes: main.c:467-477), so such a call allocates
 * nothing and needs 2 x (64 MiB + m) of device memory whatever the text's length.  The workspaces stay allocated
 * between calls; this frees them (call it with no *_count_host / legacy call in flight). */
void smh_host_path_release(void);
/* The legacy GPU names of smatcher.h (cuda_ac1..5, cuda_wm1..5, search_wu, search_wu2) take the caller's tables with every call; the
 * library keeps the handle it compiled from them (and its table set on the device) until a call arrives with other pointers,
 * shapes or contents -- main.c:583-592,623-648 runs the five variants back to back on the same tables: one compile, one upload.
 * smh_host_path_release() frees the kept handles as well.  This counts the compiles so far (tests, bench.py `preproc`). */
uint64_t smh_legacy_handle_builds(void);
/* Piece size of that path in bytes (rounded down to 4 KiB, at least 4 KiB); 0 = back to the default.  Returns the size in
 * force before the call.  Tests shrink the pieces so that a small text has hundreds of piece boundaries; counts never depend
 * on it.  Takes effect for calls that start afterwards. */
uint64_t smh_host_path_set_piece(uint64_t bytes);

/* ---- Environment.  The library reads exactly these variables -- never on the path of a scan or a launch -- and none of them can
 * change a count:
 *   SMH_ADAPT=0                 read once per process: handles keep the engine their compile chose (no adaptive switching, no
 *                               reporting launches)
 *   SMH_HOST_PIECE_KIB=N        read once per process: default piece size of the host-pointer path (smh_host_path_set_piece
 *                               overrides it)
 *   SMH_MULTI_SHARE_DEVICE=1    read by smh_multi_create: logical shards share the visible devices (one-card rehearsal), same
 *                               as the SMH_MULTI_SHARE_DEVICE flag
 * Nothing else is read: the development knobs of earlier rounds (SMH_WM_TUNE, SMH_AC_TUNE, SMH_HASH_TUNE, SMH_KEY_TUNE,
 * SMH_PSET_TUNE) exist only in tests/emu/libsmatcher_hip_testing.so, built with -DSMH_TESTING (csrc/smh_tune.h). ---- */

/* ---- synthetic corpus (clean-room stand-in for the reference's missing helper.c:
 *      load_files / create_multiple_pattern_with_hits, main.c:49,453) ----
 * Counter-based splitmix64: symbol i = mix(seed + (i+1)*0x9E3779B97F4A7C15) % alphabet,
 * so any slice [offset, offset+n) can be produced independently on host or device. */
uint64_t smh_splitmix64_at(uint64_t seed, uint64_t index);
void smh_corpus_text_host(unsigned char *out, uint64_t n, uint64_t offset, uint64_t seed, int alphabet);
int smh_corpus_text_device(unsigned char *d_out, uint64_t n, uint64_t offset, uint64_t seed,
                           int alphabet, void *stream);
/* p patterns of m symbols, flat; every `from_text_every`-th one is a substring of the
 * synthetic text (text_seed, n_text) so that matches exist; <= 0: all uniform random */
void smh_corpus_patterns(unsigned char *out, int m, int p_size, uint64_t seed, int alphabet,
                         uint64_t text_seed, uint64_t n_text, int from_text_every);

/* Non-uniform corpora (round 4; csrc/corpus_gen.h).  The reference's own data sets are E.coli, A.thaliana, swiss-prot
 * and world192 (main.c:39-109; not in the upstream repository): repeats, low-complexity runs, skewed symbol
 * frequencies -- i.i.d. uniform text is every filter engine's best case.  `kind`:
 *   SMH_CORPUS_UNIFORM      the text of smh_corpus_text_host / _device
 *   SMH_CORPUS_DNA_REPEATS  alphabet 4: order-3 Markov text with 15 % copies of 64 library blocks, 5 % tandem repeats,
 *                           10 % low-complexity (poly-A) runs, in 1 KiB blocks
 *   SMH_CORPUS_SKEWED       any alphabet: Zipf-like symbol frequencies (20 symbols: a protein's spread; 256: natural-
 *                           language-like, 4.3 bits per symbol), 15 % library blocks, 5 % low-complexity runs
 *   SMH_CORPUS_PLANTED      any alphabet: uniform text in which one 32-symbol word recurs in every 64-byte cell;
 *                           smh_corpus_patterns_kind puts the word's first m symbols into the set as pattern 0
 * Every slice [offset, offset + n) can be produced independently on host or device (same bytes).  Patterns that are
 * not substrings of the text are slices of an unrelated text of the same kind.  SMH_EINVAL for a kind / alphabet
 * that does not exist. */
#define SMH_CORPUS_UNIFORM 0
#define SMH_CORPUS_DNA_REPEATS 1
#define SMH_CORPUS_SKEWED 2
#define SMH_CORPUS_PLANTED 3
int smh_corpus_text_host_kind(unsigned char *out, uint64_t n, uint64_t offset, uint64_t seed, int alphabet, int kind);
int smh_corpus_text_device_kind(unsigned char *d_out, uint64_t n, uint64_t offset, uint64_t seed, int alphabet, int kind,
                                void *stream);
int smh_corpus_patterns_kind(unsigned char *out, int m, int p_size, uint64_t seed, int alphabet, uint64_t text_seed,
                             uint64_t n_text, int from_text_every, int kind);

/* ---- byte-range shards: main.c:375-378,464-477 with the true length of the last shard ---- */
void smh_shard_range(uint64_t n, int n_shards, int shard, int m, uint64_t *begin, uint64_t *end);

#define SMH_ALGO_AC 0
#define SMH_ALGO_WM 1
/* a third engine Aho-Corasick handles keep beside a text-dependent plan or engine (round 4): the automaton as PLAIN stride-1
 * images with K = m in LDS -- one lookup per symbol whatever the text, no verify stage -- the whole set in one image when
 * it fits, else the patterns cut into up to 16 runs whose own automata fit, scanned one after the other
 * (smh_ac_info.flat_parts).  A hybrid image's speed depends on how deep the text keeps its lanes in the trie, a filter's on
 * how many columns survive it; this engine's on nothing.  Only smh_*_set_scan_engine and smh_adapt_info name it. */
#define SMH_ENGINE_AC_FLAT 2
/* a fourth engine (round 5), held by Aho-Corasick and Wu-Manber handles alike: the KEY engine.  All patterns of a run have
 * ONE length m (smatcher.h:89-106), so the window that ends at a column is a number of m * bits bits and the set is a set
 * of such numbers: it lives in LDS as a two-table cuckoo hash of the keys themselves and every column is one exact
 * membership test -- two independent LDS reads, two compares -- in ONE pass, with no verify stage and a rate that depends
 * neither on the text nor on the patterns.  Taken by sets with m * bits <= 64 (alphabet 4: m <= 32; 20 letters: m <= 12;
 * bytes: m <= 8) whose keys fit 156 KiB of LDS at half load (about 18 000 keys of 32 bits, 9 000 of 64);
 * smh_*_info.key_slots says whether the handle holds it. */
#define SMH_ENGINE_KEYS 3
/* a fifth engine (round 5), held by Wu-Manber handles over byte-like alphabets whose set the key engine does not take (too many
 * patterns, m * bits > 64): the WINDOW-HASH engine.  A Bloom filter of the rolling hash of the WHOLE m-byte window in LDS -- its
 * false-positive rate is the table's load (about 4 % at 100 000 patterns) whatever the text, where a q-gram filter passes every
 * column whose grams are common in the text -- and the patterns themselves in a two-table cuckoo hash in device memory: a
 * surviving column costs two dependent round trips (window, both slots), true match or not.  4 <= m <= 32. */
#define SMH_ENGINE_HASH 4
#define SMH_ENGINES 5

/* ---- Aho-Corasick ---- */
typedef struct smh_ac smh_ac;

typedef struct smh_ac_info {
    uint32_t alphabet;
    uint32_t m;            /* pattern length the automaton was compiled for */
    uint32_t states;       /* reachable states in the reference numbering (== ac_table.idcounter) */
    uint32_t finals;       /* accepting states (== ac_table.patterncounter) */
    uint32_t rows;         /* DFA rows kept on the device (accepting leaves are folded away) */
    uint32_t entry_bytes;  /* 2 or 4 */
    uint32_t lds_rows;     /* rows of the depth-K scan automaton staged in LDS by the tuned kernel */
    uint32_t lds_bytes;
    uint64_t table_bytes;  /* full DFA size in HBM */
    uint32_t scan_depth;   /* K: the LDS automaton is the Aho-Corasick machine of the K-symbol prefixes */
    uint32_t scan_stride;  /* text symbols consumed per LDS lookup (1 or 2) */
    uint32_t scan_exact;   /* 1: K == m, a flagged transition is a match; 0: candidates are verified in HBM */
    uint32_t scan_full_rows; /* hybrid stride-2 image: rows below this id hold 16 two-symbol entries, deeper
                              * rows are 4-byte item lists; 0 for the plain stride-1 / stride-2 images */
    uint32_t scan_engine;    /* SMH_ALGO_AC: the automaton kernels scan; SMH_ALGO_WM: even the best LDS automaton
                              * would be verify-bound (alphabet-256 sets, thousands of long DNA patterns), so
                              * smh_ac_scan / smh_ac_positions run the suffix-filter kernels on the same patterns
                              * -- same count, several times faster; round 3: also a depth-cut plan (K < m) whose
                              * estimate the pair-gram filter beats (the headline's 1000 patterns of 16 / 32 symbols:
                              * 0.174 against 0.205 ms/GiB).  smh_ac_set_scan_plan with a forced stride or depth, or
                              * smh_ac_set_scan_engine(SMH_ALGO_AC), switches back to the automaton kernels; (0, 0) /
                              * -1 restores the choice. */
    uint32_t scan_dense;     /* 1: the dense plan scans (alphabet 4, m <= 8): the automaton completed to all 4^m strings, its
                              * state the rolling code of the last m symbols, acceptance one bit per string in LDS
                              * (two END columns per lookup); scan_stride / scan_depth describe the ordinary plan kept beside it */
    uint32_t verify_in_registers; /* scan_engine == SMH_ALGO_WM: smh_wm_info.verify_in_registers of that engine */
    uint32_t gram_kind;      /* scan_engine == SMH_ALGO_WM: smh_wm_info.gram_kind of that engine */
    uint32_t adaptive;       /* 1: the handle holds both engines and no engine or plan is forced: scan_engine is where every
                              * device starts, and from then on the engine follows what the launches report about the text
                              * (smh_ac_get_adapt) */
    uint32_t flat_parts;     /* launches of the text-independent engine (SMH_ENGINE_AC_FLAT): the set as that many exact stride-1
                              * automata that each fit LDS whole, scanned one after the other; 0: the handle keeps none */
    uint32_t key_slots;      /* round 5: slots of the key table the handle keeps (SMH_ENGINE_KEYS); 0: it keeps none */
    uint32_t hash_slots;     /* round 5: slots of the window-hash engine's pattern table (SMH_ENGINE_HASH; byte-like alphabets without a key table); 0: none */
    uint32_t reserved[4];    /* zero; library 0.2 grew this struct -- later fields come out of here (key_slots, hash_slots did) */
} smh_ac_info;

/* What the library has learned about the text it scans with a handle on the CURRENT device (round 4).  The engine that
 * serves an entry point is chosen at compile time from rates measured on pseudo-random text; a filter engine's speed,
 * though, depends on the text (every surviving column is verified one by one) -- the reference's own corpora are
 * genomes, proteins and English (main.c:39-109).  The filter kernels and the depth-cut automaton kernels therefore
 * report, per launch of 32 MiB or more, their duration on the device and the number of columns they had to verify,
 * and a handle that holds both engines switches to the other one for the NEXT launch when that is measured (or, untried
 * on this text, estimated) clearly faster; the filter kernels' verify mode follows the measured survivor rate the same
 * way.  No synchronisation: a launch that has not finished has not reported.  SMH_ADAPT=0 in the environment disables it. */
typedef struct smh_adapt_info {
    uint32_t struct_size;      /* in: sizeof(smh_adapt_info) */
    uint32_t adaptive;         /* 1: both engines at hand, none forced */
    uint32_t engine;           /* SMH_ALGO_AC / SMH_ALGO_WM / SMH_ENGINE_AC_FLAT: the kernels the next tuned scan on this device runs */
    uint32_t flips;            /* engine changes so far on this device */
    uint32_t reports;          /* launches that have reported */
    uint32_t reserved;
    double ms_per_gib[SMH_ENGINES];     /* [SMH_ALGO_AC | SMH_ALGO_WM | SMH_ENGINE_AC_FLAT]: the better of the engine's last two measured launches, 0 = not measured on this text */
    double events_per_4k[SMH_ENGINES];  /* [SMH_ALGO_AC]: candidates queued, [SMH_ALGO_WM]: surviving columns, per 4 KiB of text */
    double est_ms_per_gib[SMH_ENGINES]; /* the compile's estimates on random text; 0 = the handle has no such engine */
    double verify_density;     /* surviving columns per column the filter launcher currently plans its verify mode for (< 0: the compile's) */
} smh_adapt_info;

/* from the reference-layout tables preproc_ac filled (rows = m*p_size+1 as main.c:410-420 sizes them) */
smh_ac *smh_ac_compile_tables(const int *state_transition, const unsigned int *state_supply,
                              const unsigned int *state_final, uint64_t rows, int alphabet, int m);
/* from patterns: builds the reference tables internally, then compiles them */
smh_ac *smh_ac_compile_patterns(const unsigned char *pattern_flat, int m, int p_size, int alphabet);
int smh_ac_get_info(const smh_ac *ac, smh_ac_info *out);
int smh_ac_get_adapt(smh_ac *ac, smh_adapt_info *out);
/* tuning / test knob: rebuild the LDS scan automaton with a forced stride (1 or 2, 3 = hybrid stride 2,
 * 4 = the dense plan of smh_ac_info.scan_dense; 0 = choose) and a forced depth K (1..min(m,65); 0 = the deepest that fits; for the hybrid image
 * bits 8..15 may force the depth D of its full rows).  SMH_EUNSUP when it does not fit LDS. */
int smh_ac_set_scan_plan(smh_ac *ac, int stride, int depth);
/* test / tuning knob, the mirror of smh_wm_set_scan_engine: SMH_ALGO_AC runs the automaton kernels on the plan the handle
 * holds, SMH_ALGO_WM the suffix-filter engine (SMH_EUNSUP when the compile kept none) -- either way the adaptive choice
 * is off; -1 restores the compile's choice and the adaptive engine */
int smh_ac_set_scan_engine(smh_ac *ac, int engine);
/* asynchronous: adds the number of matches in d_text[0, n) to *d_count (device uint64).
 * d_text must be 16-byte aligned; n may exceed 2^32.
 * Threads and streams (round 5; the reference is single-threaded with global state, smatcher.h:71-73): the scan, positions and
 * count_host calls of ONE handle may be issued from several host threads and on several streams.  A handle owns per-device
 * state -- the depth-cut kernels' candidate queue, the report slots of the adaptive engine -- so the library (i) serialises the
 * host side of those calls per handle and device with a mutex and (ii) ORDERS the handle's launches on the device: a launch on
 * another stream than the handle's previous one first waits (hipStreamWaitEvent, the host does not block) for that one.  The same
 * stream as before costs nothing; the first change of stream costs one event created and recorded on the previous stream, once per handle and device (the
 * host never blocks).  Launches
 * of DIFFERENT handles are independent.  Exceptions: inside a stream capture no ordering is applied (capture a handle on one
 * stream only); smh_pset handles and SMH_ADAPT=0 processes keep the old rule -- scans of the same handle must not overlap.
 * smh_*_free must not run beside any other call on the handle.
 * The FIRST tuned scan of a handle on a device is not asynchronous: it uploads
 * the handle's tables with blocking copies, and -- a handle with more than one engine (smh_ac_info.adaptive), a
 * text of 1 GiB or more, no stream capture in progress -- it scans the first 256 MiB and waits for that launch's
 * report before it commits the rest of the text to an engine (DESIGN.md 3.4 "first look"; SMH_ADAPT=0 turns it off). */
int smh_ac_scan(smh_ac *ac, const unsigned char *d_text, uint64_t n, uint64_t *d_count,
                int variant, void *stream);
/* match positions (the reference only has commented-out printf's, ac/ac.c:217):
 * appends the END column of every match in d_text[0, n) to d_positions[*d_cursor ...] (device
 * uint64 array of `capacity` entries) and advances the device counter *d_cursor by the number of
 * matches.  Output order is unspecified.  Entries that do not fit are dropped but still counted:
 * *d_cursor > capacity afterwards means "call again with a bigger buffer". */
int smh_ac_positions(smh_ac *ac, const unsigned char *d_text, uint64_t n, uint64_t *d_positions,
                     uint64_t capacity, uint64_t *d_cursor, void *stream);
/* blocking convenience: upload host text, scan, return count and kernel-only seconds */
int smh_ac_count_host(smh_ac *ac, const unsigned char *text, uint64_t n, int variant,
                      uint64_t *count, double *kernel_seconds);
void smh_ac_free(smh_ac *ac);

/* ---- Wu-Manber ---- */
typedef struct smh_wm smh_wm;

typedef struct smh_wm_info {
    uint32_t alphabet;
    uint32_t m;
    uint32_t patterns;        /* as given */
    uint32_t distinct;        /* after de-duplication */
    uint32_t shiftsize;       /* reference SHIFT table length (wu/wu.c:18-47) */
    uint32_t shift_zero;      /* reference SHIFT entries equal to 0 */
    uint32_t block_symbols;   /* device block size (symbols hashed by the LDS filter) */
    uint32_t filter_log2;     /* log2 of the LDS filter's bit count */
    uint32_t filter_exact;    /* 1: a filter hit IS a match (block == whole pattern, direct index) */
    uint32_t filter_hashed;   /* 1: block code is hashed into the filter, 0: indexes it directly */
    uint32_t verify_slots;    /* open-addressing slots of the HBM verify table (0 when exact) */
    uint32_t lds_bytes;
    uint32_t scan_engine;     /* SMH_ALGO_WM: this path's kernels scan; SMH_ALGO_AC: a small-alphabet set of long
                               * patterns whose automaton fits LDS is scanned by the automaton kernels (faster
                               * than a non-exact direct filter; same count) -- see smh_wm_set_scan_engine */
    uint32_t gram_planes;     /* > 0: the scan runs the q-gram shift-or filter with this many positional planes
                               * (one lookup per column, or per two columns on the 4-letter alphabet) */
    uint32_t verify_in_registers; /* the verify stage a launch takes on text like the compile's.  1: pair-like form (4-letter alphabet) with
                               * next to no surviving columns (under 0.25 per 4 KiB) and m <= 33: a survivor's window is hashed by its
                               * own lane out of the text registers (kernel instance wm_gram_kernel<1 | 5, ., 5 | 6, false>).  2 (round
                               * 6): the 4-letter forms with up to 40 survivors per 4 KiB: survivors queued across chunks, their windows
                               * re-read from L2, probes pipelined (wm_gram_kernel<1 | 3 | 5, ., 3 | 4, false>).  0: a staged LDS copy
                               * of the chunk, or the form's own stage (byte forms) */
    uint32_t gram_kind;       /* form of the q-gram filter (== the kernels' KIND template value): 0 none, 1 symbol pairs (7-symbol
                               * grams, two columns per lookup), 2 hashed byte grams (one plane per offset), 3 8-symbol grams, 5 8-symbol
                               * grams at two columns per lookup, 6 flat byte grams (one Bloom set for all offsets); round 6: 8 / 9 = forms 2 / 6 in a 143.9 KiB
                               * table, 11 = the flat set over four-byte grams (alphabets whose three-symbol grams a set saturates) */
    uint32_t adaptive;        /* as smh_ac_info.adaptive: this handle also holds an automaton engine and follows the launches' reports */
    uint32_t key_slots;       /* round 5: as smh_ac_info.key_slots */
    uint32_t hash_slots;      /* round 5: slots of the window-hash engine's pattern table (SMH_ENGINE_HASH); 0: the handle keeps none */
    uint32_t verify_ck_slots; /* round 5: > 0: the verify entries also exist as a cuckoo hash of that many slots (sets whose bucket table
                               * exceeds 512 KiB), which the pipelined probes of the filter kernels read */
    uint32_t reserved[4];
} smh_wm_info;

/* from patterns; the reference-layout SHIFT / PREFIX tables are built internally */
smh_wm *smh_wm_compile(const unsigned char *pattern_flat, int m, int p_size, int alphabet);
/* from the caller's reference-layout tables (as preproc_wu / preproc_wu2 filled them) */
smh_wm *smh_wm_compile_tables(const unsigned char *pattern_flat, int m, int p_size, int alphabet,
                              const int *SHIFT, const int *PREFIX_value, const int *PREFIX_index,
                              const int *PREFIX_size);
int smh_wm_get_info(const smh_wm *wm, smh_wm_info *out);
int smh_wm_get_adapt(smh_wm *wm, smh_adapt_info *out);
/* test / tuning knob: SMH_ALGO_WM forces this path's own kernels, SMH_ALGO_AC the automaton engine
 * (SMH_EUNSUP when the set has none), -1 restores the choice made at compile time */
int smh_wm_set_scan_engine(smh_wm *wm, int engine);
int smh_wm_scan(smh_wm *wm, const unsigned char *d_text, uint64_t n, uint64_t *d_count,
                int variant, void *stream);
/* END columns of all matches (wu/wu.c:93 printed them from commented-out code); same contract as
 * smh_ac_positions */
int smh_wm_positions(smh_wm *wm, const unsigned char *d_text, uint64_t n, uint64_t *d_positions,
                     uint64_t capacity, uint64_t *d_cursor, void *stream);
int smh_wm_count_host(smh_wm *wm, const unsigned char *text, uint64_t n, int variant,
                      uint64_t *count, double *kernel_seconds);
void smh_wm_free(smh_wm *wm);

/* ---- SOG (sog/sog8.c, cuda/cuda_sog.cu): shift-or over 3-grams, patterns of length 8 ----
 * SMH_VARIANT_TABLE walks the caller's tables as given -- T8 from HBM with the shift-or state per lane, then the
 * 2-level bitmap, the binary search over the sorted hashes and the 8-byte compare (sog/sog8.c:51-115);
 * SMH_VARIANT_TUNED scans the same patterns with the tuned Wu-Manber kernels.  Both return the number of 8-byte
 * windows that equal a pattern.  (The reference's own count is not a function of its inputs: sog/sog8.c:135 sets
 * the bitmap from an uninitialised variable; preproc_sog8 in smatcher.h documents the defined contents.) */
typedef struct smh_sog smh_sog;
smh_sog *smh_sog_compile_tables(const uint8_t *T8, const uint32_t *scanner_hs, const int *scanner_index,
                                const uint8_t *scanner_hs2, const unsigned char *pattern_flat, int p_size);
int smh_sog_scan(smh_sog *sg, const unsigned char *d_text, uint64_t n, uint64_t *d_count, int variant, void *stream);
int smh_sog_count_host(smh_sog *sg, const unsigned char *text, uint64_t n, int variant, uint64_t *count, double *kernel_seconds);
void smh_sog_free(smh_sog *sg);

/* ---- one process, several GPUs: the reference driver's MPI layer for this path (main.c:464-489, 654-657) ----
 * The text is split into byte ranges, one per device, each with `halo` bytes of the next range behind it
 * (main.c:467-477; every scan uses the TRUE length of its range); a count call launches the tuned kernel on
 * every device side by side and adds the 64-bit counts with ONE ncclAllReduce(ncclUint64, ncclSum) over an
 * RCCL communicator of the devices (ncclCommInitAll) -- the MPI_Reduce of main.c:656 over xGMI.  Handles are
 * shared: a compiled automaton keeps one table set per device.
 * Threading contract of the whole library: a compiled handle (smh_ac, smh_wm, ...) may be scanned from several
 * host threads at once, each with its own current device -- the per-device table sets are published under a
 * mutex and built outside it; compiling, re-planning and freeing a handle must not run beside a scan of it.  An
 * smh_multi object is driven from ONE host thread (it starts one thread per device itself where that pays). */
typedef struct smh_multi smh_multi;
#define SMH_MULTI_MAX_DEVICES 16
#define SMH_MULTI_HOST_SUM 1 /* smh_multi_create: when RCCL cannot be loaded, add the counts on the host instead of failing */
#define SMH_MULTI_NO_RCCL 2  /* never create a communicator (host sum); for comparison runs */
#define SMH_MULTI_SHARE_DEVICE 4 /* one-card rehearsal of the N-device flow: `devices` may name a device more than once (NULL:
                                  * logical shard i sits on device i mod the visible ones); every logical shard keeps a stream,
                                  * a text range, a counter and table sets of its own, the counts are added on the host (an RCCL
                                  * communicator takes a device once).  Also switched on by SMH_MULTI_SHARE_DEVICE=1 in the
                                  * environment, so that `smatcher -ranks 4` and bench.py's native leg can be rehearsed as they are */
/* devices == NULL: devices 0 .. n_devices-1 */
int smh_multi_create(smh_multi **out, const int *devices, int n_devices, int flags);
int smh_multi_device_count(const smh_multi *mg);
int smh_multi_uses_rccl(const smh_multi *mg);
/* place a text: copied from the host, or the synthetic corpus generated range by range on its own device;
 * later scans need pattern length - 1 <= halo */
int smh_multi_load_text(smh_multi *mg, const unsigned char *text, uint64_t n, int halo);
int smh_multi_generate_text(smh_multi *mg, uint64_t n_total, uint64_t seed, int alphabet, int halo);
/* *total = the all-reduced count (checked against the sum of the per-device counts), per_device[i] (optional,
 * smh_multi_device_count entries) = device i's own count, *seconds (optional) = wall time of launches + reduce */
int smh_multi_ac_count(smh_multi *mg, smh_ac *ac, uint64_t *total, uint64_t *per_device, double *seconds);
int smh_multi_wm_count(smh_multi *mg, smh_wm *wm, uint64_t *total, uint64_t *per_device, double *seconds);
/* build the handle's table set on every device of `mg` side by side (one host thread per device) and run its kernel
 * once per device on a few KiB, so that the first count call measures what the tenth does: launches + reduce.  The
 * count calls do the same before their clock starts; this only moves it out of the first call. */
int smh_multi_ac_prepare(smh_multi *mg, smh_ac *ac);
int smh_multi_wm_prepare(smh_multi *mg, smh_wm *wm);
void smh_multi_free(smh_multi *mg);

/* ---- Set-Horspool (sh/sh.c, cuda/cuda_sh.cu) ----
 * The reversed trie of the patterns plus a bad-character table.  SMH_VARIANT_TABLE walks the
 * reference-layout trie as given with the caller's bmBc driving a per-lane skip loop;
 * SMH_VARIANT_TUNED scans the patterns read back from the trie with the tuned Wu-Manber kernels (the
 * first levels of a reversed trie are a suffix-block filter, the rest its verify stage) or, for sets
 * Wu-Manber cannot take (m < 3, other alphabets), the automaton kernels.  Both return the number of
 * end columns of pattern occurrences, which is what search_sh counts for a valid bmBc. */
typedef struct smh_sh smh_sh;
typedef struct smh_sh_info {
    uint32_t alphabet;
    uint32_t m;
    uint32_t states;        /* == struct ac_table.idcounter of preproc_sh */
    uint32_t finals;        /* == patterncounter */
    uint32_t tuned_engine;  /* SMH_ALGO_WM or SMH_ALGO_AC */
    uint32_t reserved[3];
} smh_sh_info;
/* from the reference-layout tables preproc_sh filled (rows = m*p_size+1 as main.c:410-420 sizes them) */
smh_sh *smh_sh_compile_tables(const int *state_transition, const unsigned int *state_final, uint64_t rows,
                              int alphabet, int m);
smh_sh *smh_sh_compile_patterns(const unsigned char *pattern_flat, int m, int p_size, int alphabet);
int smh_sh_get_info(const smh_sh *sh, smh_sh_info *out);
/* the valid set-Horspool table of the handle's patterns: bmBc[alphabet] */
int smh_sh_valid_bmbc(const smh_sh *sh, int *bmBc);
/* asynchronous, same contract as smh_ac_scan.  bmBc (host, alphabet ints) may be NULL = the valid table;
 * an entry outside 1..valid is refused (SMH_EINVAL): the reference would skip matches with it */
int smh_sh_scan(smh_sh *sh, const unsigned char *d_text, uint64_t n, const int *bmBc, uint64_t *d_count,
                int variant, void *stream);
int smh_sh_count_host(smh_sh *sh, const unsigned char *text, uint64_t n, const int *bmBc, int variant,
                      uint64_t *count, double *kernel_seconds);
void smh_sh_free(smh_sh *sh);

/* ---- Set Backward Oracle Matching (sbom/sbom.c, cuda/cuda_sbom.cu) ----
 * The factor oracle of the reversed patterns with its per-state pattern lists.  SMH_VARIANT_TABLE runs
 * the reference's loop per lane over the tables as given (walk, compare the listed patterns, skip
 * max(m - j, 1)); SMH_VARIANT_TUNED scans the patterns with the tuned Wu-Manber kernels (the oracle is
 * a suffix-first filter followed by a compare, which is what those kernels do) or the automaton
 * kernels for sets Wu-Manber cannot take.  Both return the number of end columns of occurrences. */
typedef struct smh_sbom smh_sbom;
typedef struct smh_sbom_info {
    uint32_t alphabet;
    uint32_t m;
    uint32_t states;        /* == struct sbom_table.idcounter */
    uint32_t patterns;      /* == patterncounter (duplicates included) */
    uint32_t listed;        /* entries over all per-state lists */
    uint32_t tuned_engine;  /* SMH_ALGO_WM or SMH_ALGO_AC */
    uint32_t reserved[2];
} smh_sbom_info;
/* from the tables preproc_sbom filled (rows = m*p_size+1; state_final_multi has rows*200 entries) */
smh_sbom *smh_sbom_compile_tables(const unsigned char *pattern_flat, int m, int p_size, int alphabet,
                                  const int *state_transition, const unsigned int *state_final_multi, uint64_t rows);
smh_sbom *smh_sbom_compile_patterns(const unsigned char *pattern_flat, int m, int p_size, int alphabet);
int smh_sbom_get_info(const smh_sbom *sb, smh_sbom_info *out);
int smh_sbom_scan(smh_sbom *sb, const unsigned char *d_text, uint64_t n, uint64_t *d_count, int variant, void *stream);
int smh_sbom_count_host(smh_sbom *sb, const unsigned char *text, uint64_t n, int variant, uint64_t *count,
                        double *kernel_seconds);
void smh_sbom_free(smh_sbom *sb);

/* ---- the key engine by itself (round 5; what SMH_ENGINE_KEYS runs inside smh_ac / smh_wm handles) ----
 * count = |{ e in [m-1, n) : text[e-m+1 .. e] in set(patterns) }|, the quantity of search_ac / search_wu (ac/ac.c:198-222,
 * wu/wu.c:49-107), by an exact hash-set lookup per column.  NULL (smh_last_error says why) when the set is not one the
 * engine takes: m * ceil(log2 alphabet) > 64, or more distinct patterns than two tables in LDS hold. */
typedef struct smh_keys smh_keys;
typedef struct smh_keys_info {
    uint32_t struct_size;  /* in: sizeof(smh_keys_info) */
    uint32_t alphabet, m;
    uint32_t keys;         /* distinct patterns */
    uint32_t key_bits;     /* m * bits per symbol */
    uint32_t slot_bytes;   /* 4 or 8 */
    uint32_t slots;        /* per table (two tables) */
    uint32_t lds_bytes;    /* the image */
    double est_ms_per_gib;
    /* round 6 (struct_size 48; a caller compiled against the 40-byte struct of round 5 gets the fields above only) */
    uint32_t layout;       /* 0 = two-table cuckoo hash: two LDS reads per column; 1 = bucket image: one 8-byte read per column, `slots` = the
                            * two-slot buckets of the primary table, keys of buckets with three or more in a small overflow table (csrc/key_hash.h) */
    uint32_t overflow_keys;
} smh_keys_info;
smh_keys *smh_keys_compile_patterns(const unsigned char *pattern_flat, int m, int p_size, int alphabet);
int smh_keys_get_info(const smh_keys *k, smh_keys_info *out);
/* asynchronous, same contract as smh_ac_scan (16-byte aligned device text, *d_count is added to) */
int smh_keys_scan(smh_keys *k, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream);
int smh_keys_positions(smh_keys *k, const unsigned char *d_text, uint64_t n, uint64_t *d_positions, uint64_t capacity,
                       uint64_t *d_cursor, void *stream);
void smh_keys_free(smh_keys *k);

/* ---- pattern sets with mixed lengths ----
 * The reference API carries ONE pattern length per run (preproc_ac / preproc_wu take a single m;
 * feeding ac_addstring mixed lengths marks wrong states final, ac/ac.c:136-143,183-186), so the
 * bit-exact meaning of a mixed set is the length-class decomposition: group the patterns by
 * length, run the reference once per length, sum the counts -- the number of (end column, distinct
 * pattern) occurrences.  A set handle does exactly that on the device: one compiled automaton
 * (SMH_ALGO_AC) or Wu-Manber table set (SMH_ALGO_WM) per distinct length, all scanning the same
 * resident text on the caller's stream and adding into the same counter.  Wu-Manber needs m >= 3
 * (wu/wu.c:119-125): classes of length 1 and 2 of a SMH_ALGO_WM set are compiled as automata. */
typedef struct smh_pset smh_pset;

typedef struct smh_pset_info {
    uint32_t alphabet;
    uint32_t algorithm;
    uint32_t classes;      /* distinct pattern lengths */
    uint32_t patterns;     /* as given */
    uint32_t min_length;
    uint32_t max_length;
    uint32_t one_pass;     /* 1: the text is read ONCE.  SMH_ALGO_AC sets: one automaton whose states carry joined
                            * (suffix-closed) output counts, cut at the deepest level that fits LDS, longer patterns
                            * verified along the goto trie.  SMH_ALGO_WM sets with 2..32 lengths, all >= 3: a filter
                            * proposes END columns, each survivor is verified per length class -- on the 4-letter
                            * alphabet a q-gram shift-or filter over the FULL patterns (two plane groups: patterns of
                            * 14 symbols and more / the shorter ones) while candidates stay below one column in 2000,
                            * else a block filter over the patterns' last min-length symbols while it passes < 0.4 %
                            * of the columns; sets neither serves run the automaton above.  0: one scan per class. */
    uint32_t passes;       /* scans of the text one smh_pset_scan makes: 1 (one_pass), 2 (SMH_ALGO_WM sets on the 4-letter
                            * alphabet that neither single pass serves: the patterns of 14 symbols and more through the
                            * q-gram filter, the shorter ones through the automaton), else the number of classes */
} smh_pset_info;

/* patterns: the p_size patterns back to back (pattern j has lengths[j] symbols, each < alphabet) */
smh_pset *smh_pset_compile(const unsigned char *patterns, const uint32_t *lengths, int p_size, int alphabet,
                         int algorithm);
int smh_pset_get_info(const smh_pset *set, smh_pset_info *out);
/* class i in ascending length order: its length and its number of patterns */
int smh_pset_get_class(const smh_pset *set, uint32_t i, uint32_t *length, uint32_t *patterns);
/* asynchronous, same contract as smh_ac_scan: adds sum over classes of the class's match count */
int smh_pset_scan(smh_pset *set, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream);
/* END columns of all (column, pattern) occurrences; a column appears once per class that matches there */
int smh_pset_positions(smh_pset *set, const unsigned char *d_text, uint64_t n, uint64_t *d_positions,
                      uint64_t capacity, uint64_t *d_cursor, void *stream);
int smh_pset_count_host(smh_pset *set, const unsigned char *text, uint64_t n, uint64_t *count,
                       double *kernel_seconds);
void smh_pset_free(smh_pset *set);

#ifdef __cplusplus
}
#endif
#endif
