/*
 * csrc/smh_internal.h -- layouts shared by the C host code (ac_host.c, wm_host.c,
 * corpus.c) and the HIP side (smh_runtime.hip, *_kernels.hip).  Not installed.
 */
#ifndef SMH_INTERNAL_H
#define SMH_INTERNAL_H

#include <stdint.h>
#include <stddef.h>
#include "../../include/smatcher.h"
#include "../../include/smatcher_hip.h"
#include "smh_tune.h" /* development knobs: constants unless -DSMH_TESTING */

#ifdef __cplusplus
extern "C" {
#endif

#define SMH_MAGIC_AC 0x41434446u /* "ACDF" */
#define SMH_MAGIC_WM 0x574d424cu /* "WMBL" */
#define SMH_MAGIC_SH 0x53485452u /* "SHTR" */
#define SMH_MAGIC_SBOM 0x53424f4du /* "SBOM" */

void smh_set_error(const char *fmt, ...);
uint64_t smh_handle_serial(void); /* ac_host.c: 1, 2, 3, ... */

/* ------------------------------------------------------------------ AC
 * Device automaton (DESIGN.md "AC layout").
 *
 * full DFA   : one row per kept state, `alphabet` entries per row, entry = next
 *              row | FLAG.  FLAG (top bit) says the reference automaton is in an
 *              accepting state after this transition.  Accepting leaves have no
 *              row: a transition into leaf f is stored as (row of supply*(f)) |
 *              FLAG, exact because a leaf's goto is undefined for every symbol
 *              and the walk continues from its supply state (ac/ac.c:209-211).
 *              Rows are numbered breadth-first: row id order == depth order.
 *              Lives in HBM; only the verify stage reads it.
 * scan table : the same automaton cut at depth K <= m (the Aho-Corasick machine
 *              of the patterns' K-symbol prefixes), small enough to sit in LDS
 *              whole.  Its rows are rows [0, rows_k) of the full DFA; an entry
 *              that would leave depth K is bent to the supply state.  FLAG now
 *              means "a K-symbol pattern prefix ends here".  K == m: FLAG is a
 *              match and there is no verify stage.  K < m: the position is a
 *              candidate; it is queued with its depth-K row and later walked
 *              down the goto edges of the full DFA for the remaining m-K symbols.
 *              stride 1: entry = next row | FLAG            (alphabet entries/row)
 *              stride 2: entry = row after TWO symbols | F1<<14 | F2<<15
 *                        (alphabet^2 entries/row, 16-bit, alphabet 4 only)
 */
struct smh_ac_dev; /* opaque to C: device buffers, owned by smh_runtime.hip */
struct smh_keys;   /* key engine, below */
struct smh_hashes; /* window-hash engine, below */

struct smh_ac {
    uint32_t magic;
    int alphabet;
    int m;
    uint32_t states;      /* reachable states, reference numbering */
    uint32_t finals;      /* accepting states */
    uint32_t rows;        /* kept states == DFA rows */
    int entry_bytes;      /* 2: FLAG = 0x8000, rows <= 32768; 4: FLAG = 0x80000000 */
    void *table;          /* rows * alphabet entries, host copy */
    uint64_t table_bytes;
    int max_depth;        /* depth of the deepest kept row */
    uint32_t *depth_first; /* [max_depth + 2]: first row with depth >= d; [max_depth+1] = rows */
    int fixed_length_ok;  /* every accepting state is a leaf at depth m (chunked scans are exact) */
    uint8_t *row_depth;   /* [rows] */
    uint32_t *row_fail;   /* [rows] supply state of each kept row, as a row id */
    /* scan table (LDS image) and its plan */
    int scan_depth;       /* K */
    int scan_stride;      /* 1 or 2 symbols per lookup */
    int scan_exact;       /* K == m */
    uint32_t scan_rows;   /* rows with depth <= K */
    uint32_t scan_full_rows; /* hybrid stride-2 image: rows below this id have 16 entries, the others are
                              * compact item lists (ac_host.c hyb_build); 0 = not a hybrid image */
    int scan_entry_bytes; /* 2 or 4 (stride 2: always 2) */
    void *scan_table;     /* scan_rows * alphabet^stride entries */
    uint32_t scan_bytes;  /* padded to 16 */
    double scan_candidate_rate; /* expected candidates per text byte on uniform text (0 when exact) */
    double scan_cost;     /* plan cost model's estimate, 1.0 = an exact stride-1 scan (ac_host.c) */
    /* dense plan (alphabet 4, 3 <= m <= 8; stride code 4): the automaton completed to EVERY string of up to m symbols -- its
     * state is then the last m text symbols themselves, kept as a rolling code in a register, and the transition
     * function degenerates to "is this code accepting?" -- one bit per m-symbol string.  dense_pair is that bit set laid
     * out for two END columns per lookup (index = the code of nine symbols; as smh_wm.pair_table), dense_filter the plain
     * one (2^(2m) bits) for the bounds-checked path.  Chosen when the stride-2 image of the ordinary automaton does not
     * fit LDS: every level the patterns fill completely (8000 random 8-mers: all 4^d states down to d = 6) costs rows
     * there and nothing here. */
    int scan_dense;
    uint32_t *dense_pair;   /* 16384 dwords = 64 KiB */
    uint32_t *dense_filter; /* max(1, 4^m / 32) dwords */
    struct smh_wm *alt_wm; /* suffix-filter engine for sets whose best automaton plan is verify-bound or estimated slower, else NULL */
    int alt_off;          /* a scan plan was forced: scans use the automaton kernels regardless (== engine_forced == SMH_ALGO_AC) */
    /* round 4: the engine is a property of the handle AND the text.  flex_wm = the suffix-filter engine of a depth-cut plan
     * whichever engine the compile preferred (the same handle as hv_wm): with both engines at hand the runtime follows what
     * the launches report about the text (smh_runtime.hip "adaptive engine").  engine_forced: -1 = let it, else the engine
     * smh_ac_set_scan_engine / a forced plan named */
    struct smh_wm *flex_wm;
    struct smh_ac *flat_ac; /* the engine whose speed does not depend on the text: the set as flat_parts exact stride-1 automata that each fit
                             * LDS whole (ac_host.c, end of the compile), the first of them; NULL when the plan itself is of that kind or
                             * more than SMH_FLAT_MAX_PARTS would be needed */
    struct smh_ac *flat_next; /* in a part: the next part */
    int flat_parts;
    struct smh_keys *keys;  /* round 5: the key engine over the same patterns (key_host.c), NULL = the set is not one it takes or the
                             * handle's plan is an exact one-launch plan already */
    int engine_forced;
    uint32_t generation;  /* bumped by every re-plan / forced engine: what was prepared or warmed for the handle before is stale */
    uint64_t serial;      /* unique per compiled handle (smh_handle_serial): an address can be reused after a free, a serial cannot */
    struct smh_adapt_dev *adapt; /* per device: stats block, measurements, current engine (smh_runtime.hip) */
    struct smh_wm *hv_wm; /* verify table + patterns for the automaton kernels' verify stage (hash the window, probe), else NULL */
    /* stride-1 depth-K table in HBM: the slow path and the resolution of stride-2 "first symbol"
     * candidates read it; identical to scan_table when scan_stride == 1 */
    void *trunc1_table;
    int trunc1_entry_bytes;
    uint64_t trunc1_bytes;
    /* reference-layout tables truncated to `states` rows, for SMH_VARIANT_TABLE */
    int32_t *g_transition; /* states * alphabet, -1 = no edge, row 0 as ac_init leaves it */
    uint32_t *g_supply;
    uint32_t *g_final;
    struct smh_ac_dev *dev;
};

/* private wrapper handed out by preproc_ac: the public struct first, so a
 * struct ac_table* from the caller can be cast back */
struct smh_ac_table_box {
    struct ac_table pub;
    uint32_t magic;
    struct smh_ac *ac;
    struct ac_state root;
};

/* host-side builders (ac_host.c) */
enum { SMH_AC_REF_NONE = 0, SMH_AC_REF_COPY = 1, SMH_AC_REF_ADOPT = 2 };
struct smh_ac *smh_ac_compile_tables_impl(const int *state_transition, const unsigned int *state_supply,
                                          const unsigned int *state_final, uint64_t rows,
                                          int alphabet, int m, int ref_mode);
void smh_ac_host_free(struct smh_ac *ac);
/* choose K / stride for an LDS budget and build scan_table (+ trunc1_table); force_stride 0 = auto */
int smh_ac_plan_scan(struct smh_ac *ac, uint32_t lds_budget, int force_stride, int force_depth);
#define SMH_AC_LDS_BUDGET (160u * 1024u - 512u)
#define SMH_FLAT_MAX_PARTS 16 /* passes over the text the text-independent engine may take (0.27-0.29 ms/GiB each) */
/* engine choices compile a handle of the OTHER kind; while one is being built no further one is (an
 * automaton handle built as a Wu-Manber handle's engine must not build a Wu-Manber engine of its own) */
#ifdef __cplusplus
extern thread_local int smh_alt_engine_depth;
#else
extern _Thread_local int smh_alt_engine_depth;
#endif
#define SMH_WM_FLEX_ENGINE_COST 1.5 /* automaton plan cost up to which a Wu-Manber handle keeps the automaton as its second engine */
#define SMH_WM_ALT_ENGINE_COST 1.15 /* automaton plan cost (1.0 = 3.5 TB/s) below which it beats the non-exact direct filter */
#ifndef SMH_HYB_COMPACT0
#define SMH_HYB_COMPACT0 0x8000u /* hybrid stride-2 image: id of the first item slot of the compact part (== lane_common.h) */
#endif
#define SMH_AC_ALT_ENGINE_COST 2.5 /* above this plan cost (< ~1.4 TB/s) the suffix-filter engine scans the set */
#ifndef SMH_REGV_MAX_PER_CHUNK
#define SMH_REGV_MAX_PER_CHUNK 8.0 /* == lane_common.h; surviving columns per 4 KiB wave-chunk up to which the pair-gram kernels verify in registers (wm_lane.h smh_wm_regv_columns) */
#endif
#ifndef SMH_L2_MIN_PER_CHUNK
#define SMH_L2_MIN_PER_CHUNK 0.02 /* == lane_common.h; from here up (to SMH_L2_DNA_MAX_PER_CHUNK) the DNA gram forms verify through the windows-from-L2 pipeline (wm_kernels.inc launch_gram) */
#endif
#ifndef SMH_L2_MIN_PER_CHUNK_REGV
#define SMH_L2_MIN_PER_CHUNK_REGV 0.25 /* (both headers) ... for the forms that can verify in registers (pair form, two-column 8-grams): below it -- the headline sets, 0.003-0.03 per chunk -- the in-register instance is 2 % faster (no pipeline to move along) */
#endif
#ifndef SMH_L2_DNA_MAX_PER_CHUNK
#define SMH_L2_DNA_MAX_PER_CHUNK 40.0 /* (both headers) ... up to here: at 46 per chunk the staged verify measured 3 % faster again */
#endif
#ifndef SMH_REGV_WANTED
#define SMH_REGV_WANTED(per_chunk) ((per_chunk) <= SMH_REGV_MAX_PER_CHUNK) /* == lane_common.h */
#endif
#define SMH_AC_ALT_ENGINE_MARGIN_MS 0.01 /* a depth-cut plan hands the scan to the gram filter when that is estimated this much faster (ms/GiB) */
#define SMH_AC_MAX_SCAN_DEPTH 65 /* fast paths cover a halo of K - 1 <= 64 bytes */
void smh_ac_dev_free(struct smh_ac_dev *dev); /* smh_runtime.hip */
struct smh_adapt_dev;
void smh_adapt_dev_free(struct smh_adapt_dev *list); /* smh_runtime.hip */
struct smh_hashes *smh_ac_hash_engine(const struct smh_ac *ac); /* ac_host.c: what the handle runs as SMH_ENGINE_HASH, or NULL */
double smh_ac_plan_ms(const struct smh_ac *ac); /* ac_host.c: the plan model's estimate for the automaton kernels, ms per GiB */
int smh_ac_prepare_device(struct smh_ac *ac); /* smh_runtime.hip: table set of the current device, no launch */

/* ------------------------------------------------------------------ key engine (round 5; key_hash.h, key_host.c, key_lane.h)
 * The distinct patterns of one length as a two-table cuckoo hash of their keys in LDS: one exact membership test per text
 * column, no verify stage, a rate that depends neither on the text nor on the set.  Held by an automaton handle
 * (smh_ac.keys) and by a Wu-Manber handle (smh_wm.keys) as SMH_ENGINE_KEYS whenever the set is one it takes. */
#define SMH_MAGIC_KEYS 0x4b455953u /* "KEYS" */
#include "key_hash.h"
struct smh_keys_dev;
struct smh_keys {
    uint32_t magic;
    int alphabet;
    int m;
    uint32_t n_keys;          /* distinct patterns */
    struct smh_key_params P;
    void *image;              /* P.bytes: table 1, table 2 */
    double ms_est;            /* ms per GiB */
    struct smh_keys_dev *dev; /* per device: the image in device memory (smh_runtime.hip) */
};
#define SMH_KEYS_LDS_BUDGET (156u * 1024u)
#define SMH_KEYS_MS_NARROW 0.40 /* ms per GiB at steady state, 4-byte slots (32-bit and quotient keys): measured on MI355X whatever text and set (profiles/r05_keys) */
#define SMH_KEYS_MS_QUOT 0.46   /* quotient keys: two registers of rolling code, 4-byte slots */
#define SMH_KEYS_MS_WIDE 0.51   /* 8-byte slots */
#define SMH_KEYS_MS_BUCKET 0.285      /* round 6, bucket image (key_hash.h): one LDS read per column; window within the 32-bit image */
#define SMH_KEYS_MS_BUCKET_OLD 0.30   /* ... window longer than the image: one more instruction per column */
#define SMH_KEYS_MS_CROWDED_STEP 0.16 /* ... plus the crowded buckets' cost (key_host.c): per step of eight columns / per column with a sentinel in the wave */
#define SMH_KEYS_MS_CROWDED_COL 0.15
struct smh_keys *smh_keys_build(const unsigned char *patterns_flat, int m, int p_size, int alphabet, uint32_t lds_budget, const char **why);
void smh_keys_free(struct smh_keys *k);
int smh_keys_contains(const struct smh_keys *k, uint64_t key);
int smh_keys_symbol_bits(int alphabet);
void smh_keys_dev_free(struct smh_keys_dev *dev); /* smh_runtime.hip */

/* ------------------------------------------------------------------ window-hash engine (round 5; hash_engine.h, hash_host.c, hash_lane.h)
 * A Bloom filter of the WHOLE window's rolling hash in LDS -- a pass rate that does not depend on the text -- and the patterns
 * themselves in a two-table cuckoo hash in device memory: the filter engine for byte sets too large for the key engine, kept
 * by a Wu-Manber handle as SMH_ENGINE_HASH. */
#define SMH_MAGIC_HASHES 0x48415348u /* "HASH" */
#include "hash_engine.h"
struct smh_hash_dev;
struct smh_hashes {
    uint32_t magic;
    int m;
    uint32_t distinct;
    struct smh_hash_params P;
    uint32_t *bloom;        /* P.bloom_bytes: the LDS image */
    unsigned char *table;   /* 2 * P.slots slots of 4 * P.slot_dwords bytes */
    uint64_t table_bytes;
    double pass_rate;       /* non-matching columns the filter lets through */
    double ms_est;          /* ms per GiB on text without matches */
    struct smh_hash_dev *dev;
};
#define SMH_HASHES_MS_SCAN 0.43
#define SMH_HASHES_MS_SCAN3 0.50 /* round 6: with the third filter bit (four more vector instructions per column: +0.065 ms/GiB measured) */
#define SMH_HASHES_MS_PER_SURVIVOR 0.0020 /* ms per GiB per surviving column in 4 KiB: 100 000 patterns of 12 bytes, 213 per 4 KiB on uniform text 0.87 ms/GiB against 0.43 with the survivors dropped (profiles/r05_final/notes) */
struct smh_hashes *smh_hash_build(const unsigned char *patterns, int m, int distinct, const char **why);
void smh_hash_free(struct smh_hashes *k);
int smh_hash_filter_passes(const struct smh_hashes *k, const unsigned char *window);
int smh_hash_contains(const struct smh_hashes *k, const unsigned char *window);
void smh_hash_dev_free(struct smh_hash_dev *dev); /* smh_runtime.hip */

/* ------------------------------------------------------------------ mixed-length automaton (acm_host.c)
 * One Aho-Corasick automaton with joined (suffix-closed) output COUNTS for a set of patterns of different
 * lengths, cut at depth K for LDS; see acm_host.c for the construction and acm_lane.h for the scan. */
#define SMH_MAGIC_ACM 0x41434d58u /* "ACMX" */
struct smh_acm_dev;
struct smh_acm {
    uint32_t magic;
    int alphabet;
    int max_len;
    int K;               /* depth of the LDS automaton */
    int exact;           /* K >= max_len: the scan alone counts everything */
    uint32_t nodes;      /* trie nodes, breadth-first ids, root 0 */
    uint32_t scan_rows;  /* nodes of depth <= K == rows of the LDS table */
    int entry_bytes;     /* 2: candidate << 15 | count << 13 | row;  4: candidate << 31 | count << 24 | row */
    void *scan;          /* scan_rows * alphabet entries */
    uint32_t scan_bytes; /* padded to 16 */
    uint32_t *g_goto;    /* nodes * alphabet: child id or 0 (goto edges only) */
    uint8_t *g_final;    /* nodes: 1 when a pattern ends at the node */
    struct smh_acm_dev *dev;
};
struct smh_acm *smh_acm_compile(const unsigned char *patterns, const uint32_t *lengths, int p_size, int alphabet);
void smh_acm_free(struct smh_acm *a);
void smh_acm_dev_free(struct smh_acm_dev *dev); /* smh_runtime.hip */
int smh_acm_scan(struct smh_acm *a, const unsigned char *d_text, uint64_t n, uint64_t *d_count, void *stream);

/* ------------------------------------------------------------------ SH (Set-Horspool, sh_host.c)
 * The reversed trie as the reference lays it out (for the table-walking kernel), the patterns read
 * back from it, the valid bad-character table, and the tuned engine that scans them. */
struct smh_sh_dev; /* opaque to C: device buffers, owned by smh_runtime.hip */
struct smh_sh {
    uint32_t magic;
    int alphabet;
    int m;
    uint32_t states;        /* ids in use (== struct ac_table.idcounter of preproc_sh) */
    uint32_t finals;        /* == patterncounter */
    uint32_t n_patterns;    /* distinct patterns read back from the trie */
    int32_t *g_transition;  /* states * alphabet: row 0 uses 0, the other rows -1, for "no edge" */
    uint32_t *g_final;
    unsigned char *patterns; /* n_patterns * m */
    int32_t *valid_bmbc;    /* [alphabet] */
    struct smh_wm *wm;      /* tuned engine: exactly one of wm / ac */
    struct smh_ac *ac;
    struct smh_sh_dev *dev;
};
struct smh_sh_table_box { /* handed out by preproc_sh: the public struct first */
    struct ac_table pub;
    uint32_t magic;
    struct smh_sh *sh;
};
/* mixed-length sets in one pass (smh_runtime.hip) */
int smh_wm_scan_multi(struct smh_wm *suffix, struct smh_wm *const *classes, int n_classes, const unsigned char *d_text,
                      uint64_t n, uint64_t *d_count, void *stream);
int smh_wm_positions_multi(struct smh_wm *suffix, struct smh_wm *const *classes, int n_classes,
                           const unsigned char *d_text, uint64_t n, uint64_t *d_positions, uint64_t capacity,
                           uint64_t *d_cursor, void *stream);
#define SMH_PSET_MAX_ONE_PASS_CLASSES 32
#define SMH_PSET_ONE_PASS_DENSITY 0.004 /* fraction of columns the suffix filter lets through, above which one scan per class is faster */
void smh_sh_host_free(struct smh_sh *sh);
void smh_sh_dev_free(struct smh_sh_dev *dev); /* smh_runtime.hip */
int smh_sh_check_bmbc(const struct smh_sh *sh, const int *bmBc);

/* ------------------------------------------------------------------ SBOM (sbom_host.c)
 * The factor oracle as the reference lays it out, its per-state pattern lists packed from the
 * 200-entry rows into offsets + ids, the patterns, and the tuned engine that scans them. */
struct smh_sbom_dev;
struct smh_sbom {
    uint32_t magic;
    int alphabet;
    int m;
    uint32_t states;
    uint32_t n_patterns;   /* as given, duplicates included (== patterncounter) */
    uint32_t listed;       /* entries over all per-state lists */
    int32_t *g_transition; /* states * alphabet: trie edges and external transitions */
    uint32_t *g_final_off; /* [states + 1] */
    uint32_t *g_final_ids; /* pattern ids, in list order */
    unsigned char *patterns; /* n_patterns * m */
    struct smh_wm *wm;     /* tuned engine: exactly one of wm / ac */
    struct smh_ac *ac;
    struct smh_sbom_dev *dev;
};
struct smh_sbom_table_box { /* handed out by preproc_sbom: the public struct first */
    struct sbom_table pub;
    uint32_t magic;
    struct smh_sbom *sb;
};
void smh_sbom_host_free(struct smh_sbom *sb);
void smh_sbom_dev_free(struct smh_sbom_dev *dev); /* smh_runtime.hip */

/* ------------------------------------------------------------------ SOG (sog_host.c)
 * The caller's tables as given (T8, sorted hashes, their permutation, the 2-level bitmap), the patterns, and
 * the tuned engine that scans them. */
#define SMH_MAGIC_SOG 0x534f4738u /* "SOG8" */
struct smh_sog_dev;
struct smh_sog {
    uint32_t magic;
    uint32_t n_patterns;
    uint8_t *t8;       /* 2^24 */
    uint32_t *hs;      /* n_patterns, ascending */
    int32_t *index;    /* n_patterns */
    uint8_t *hs2;      /* 8192 */
    unsigned char *patterns; /* n_patterns * 8, the caller's order (scanner_index refers to it) */
    struct smh_wm *wm;
    struct smh_sog_dev *dev;
};
void smh_sog_dev_free(struct smh_sog_dev *dev); /* smh_runtime.hip */

/* ------------------------------------------------------------------ WM
 * Device tables (DESIGN.md "WM layout"):
 *   filter   : bit set in LDS indexed by the code of the window's last
 *              `block_symbols` symbols (bits_per_symbol each) -- the device
 *              SHIFT table: bit = 1  <=>  SHIFT_dev[block] == 0.  Either direct
 *              (index = code) or hashed (two bits of one 32-bit word, picked by a
 *              multiplicative hash of the code).
 *   verify   : open-addressing table in HBM keyed by FNV-1a of the whole window,
 *              {tag, pattern+1}; the device HASH/PREFIX stage.  Absent when the
 *              filter is exact (block == whole pattern, direct index).
 *   patterns : distinct patterns, sorted, m bytes each.
 *   legacy   : the reference tables (SHIFT, PREFIX_* as CSR) for SMH_VARIANT_TABLE.
 */
struct smh_wm_dev;

struct smh_wm {
    uint32_t magic;
    int alphabet;
    int m;
    int patterns;
    int distinct;
    int bits_per_symbol;
    /* tuned path */
    int block_symbols;
    int filter_log2;      /* bits in the filter = 1 << filter_log2 (>= 5) */
    int filter_exact;
    int filter_hashed;
    int filter_k;         /* hashed filter: bits per key (2..4), all in one 32-bit word */
    int filter_le4;       /* hashed filter keyed by the block's last four BYTES read as a little-endian dword
                           * (8-bit symbols, 4-symbol block): the scan takes it from its registers with one
                           * v_alignbyte instead of rolling a code */
    uint32_t *filter;     /* (1 << filter_log2) / 32 words */
    double filter_density; /* fraction of windows expected to pass on uniform text */
    int verify_log2;      /* slots = 1 << verify_log2; 0 slots when exact */
    uint32_t *verify;     /* 2 words per slot: tag, pattern index + 1 (0 = empty) */
    /* round 5: the same entries as a two-table cuckoo hash of two-slot buckets, 82 % full (100 000 patterns: 0.5 MB instead of the
     * 2 MiB above) -- what the pipelined probes of the byte-gram kernels read: random 16-byte probes fill 128-byte lines, and a table
     * that does not stay in L2 beside the streaming text cost them 1.6-1.95 x the algorithmic HBM traffic.  NULL for small sets. */
    uint32_t *verify_ck;  /* 4 * ck_buckets entries: bucket b of table t at 2 * (t * ck_buckets + b) */
    uint32_t ck_buckets, ck_seed;
    unsigned char *pat_sorted; /* distinct * m */
    /* pair filter (alphabet 4, m <= 8, exact): indexed by the code i of NINE consecutive symbols (18 bits, oldest
     * symbol highest).  The seven oldest symbols select the dword pair_table[i >> 4]; inside it the newest two
     * symbols (the pair, i & 15) select two adjacent bits: bit 2 * pair = "the m symbols ending at the 8th
     * symbol are a pattern", bit 2 * pair + 1 = "... ending at the 9th symbol".  One LDS lookup (ds_read_b32)
     * answers two end columns.  64 KiB. */
    uint32_t *pair_table;
    /* gram filter (q-gram shift-or; 64 or 128 KiB of LDS).  For every text column ONE table lookup yields a byte G
     * whose bit 7-j is CLEAR when "the q-gram that ends here is the q-gram that ends j symbols before the end of
     * some pattern" (plane j, j < gram_planes; the low 8 - gram_planes bits are always clear), and the lane state
     * S = (S << 1) | G has bit 7 clear exactly when the last gram_planes q-grams are in their planes in
     * order: the column is a candidate and goes to the verify table.  The idea of the reference's sog/sog8.c
     * (3-gram bit table T8, smatcher.h:77-80, shift-or state) with positional planes over the patterns' tail.
     *   SMH_GRAM_PAIR  alphabet 4: 7-symbol grams, table indexed by EIGHT consecutive symbols (16 bits), 16-bit
     *                  entries (G of the older seven symbols' column << 1) | G of the next column -- one lookup
     *                  and ONE v_lshl_or serve two columns, G up to 15 bits wide (planes); behind the 128 KiB image, 32 KiB
     *                  of per-gram values G for
     *                  the bounds-checked path (never staged in LDS)
     *   SMH_GRAM_OCT   alphabet 4: 8-symbol grams, 8-bit entries indexed by the gram, one lookup per column: for
     *                  pattern counts at which the 7-symbol planes fill up (8000 patterns: 39 % full, and
     *                  overlapping grams pass together: 0.4 % of the columns survive eight planes; the 8-symbol
     *                  planes are 12 % full and 1e-7 survive)
     *   SMH_GRAM_BYTE  8-bit symbols: 3-byte grams, index = top 17 bits of (gram as a little-endian 24-bit
     *                  number) * SMH_GRAM_MUL mod 2^32, 8-bit entries
     *   SMH_GRAM_OCT2  alphabet 4, round 3: 8-symbol grams with ONE lookup per TWO columns.  The lookup of a pair of
     *                  columns is indexed by the eight symbols that end at the pair's second column and yields a 16-bit
     *                  value with J <= 16 planes at ALL offsets: bit J-1-j CLEAR = "this 8-gram ends j symbols before the
     *                  end of some pattern".  S = (S << 2) | E chains every SECOND offset: after a lookup, bit J-1 clear =
     *                  the grams at offsets 0, 2, 4, ... are in place = candidate END at the pair's second column; bit J-2
     *                  clear = offsets 1, 3, 5, ... = candidate END at the NEXT column.  Half the tests per END column, but
     *                  of grams four times as selective and less correlated than the pair form's overlapping 7-symbol
     *                  grams (8000 patterns of 16 symbols: 1.4 instead of 4.4 surviving columns per 4 KiB; 16 000 of 20:
     *                  0.7, at half the lookups of SMH_GRAM_OCT) */
    int gram_kind;
    int gram_planes;
    int gram_jb;         /* SMH_GRAM_PAIR2: planes of the short-pattern group (0 = no such group) */
    void *gram_table;
    uint32_t gram_bytes;
    uint32_t sfx_slot_off, sfx_ent_off, sfx_pat_off; /* SMH_GRAM_PAIR2: byte offsets in gram_table of the verify stage's suffix index
                                                      * (wm_host.c smh_wm_build_gram_mixed); 0 = none */
    double gram_density; /* fraction of columns expected to reach the verify stage on uniform text */
    double gram_lane0;   /* pair form: columns per wave-chunk that only the assumption made for lane 0 lets through (wm_lane.h) */
    double scan_ms_est;  /* this path's own kernels, estimated ms per GiB (non-exact filters only; engine choice) */
    /* reference-layout tables */
    uint32_t shiftsize;
    uint32_t shift_zero;
    int32_t *l_shift;       /* shiftsize */
    uint32_t *l_bucket_off; /* shiftsize + 1 */
    int32_t *l_bucket;      /* 2 ints per entry: PREFIX_value, PREFIX_index */
    unsigned char *pat_orig; /* patterns * m, original order (PREFIX_index refers to it) */
    struct smh_wm_dev *dev;
    struct smh_ac *alt_ac; /* automaton engine for small-alphabet sets of long patterns when it is the faster one, else NULL */
    int alt_off;           /* smh_wm_set_scan_engine(SMH_ALGO_WM): scans use this path's own kernels regardless */
    struct smh_ac *flex_ac; /* round 4: the automaton engine kept at hand even when it is the slower one on random text (== alt_ac when that is set) */
    struct smh_keys *keys;  /* round 5: the key engine over the same patterns (key_host.c), NULL = not a set it takes / this path is exact */
    struct smh_hashes *hashes; /* round 5: the window-hash engine (hash_host.c) for byte-like sets the key engine does not take, else NULL */
    int engine_forced;     /* -1 = the runtime follows the launches' reports (round 4), else the engine smh_wm_set_scan_engine named */
    uint32_t generation;   /* bumped by smh_wm_set_scan_engine */
    uint64_t serial;       /* unique per compiled handle */
    struct smh_adapt_dev *adapt; /* per device: stats block, measurements, current engine / verify mode (smh_runtime.hip) */
};

struct smh_wm *smh_wm_compile_impl(const unsigned char *pattern_flat, int m, int p_size, int alphabet,
                                   const int *SHIFT, const int *PREFIX_value, const int *PREFIX_index,
                                   const int *PREFIX_size);
void smh_wm_host_free(struct smh_wm *wm);
/* grouped pair-gram filter over the FULL patterns of a mixed-length set, attached to the set's suffix handle;
 * 0 = built (gram_kind == SMH_GRAM_PAIR2), 1 = not applicable / too many candidates, -1 = out of memory */
int smh_wm_build_gram_mixed(struct smh_wm *suffix, const unsigned char *patterns, const uint32_t *lengths, int p_size);
/* part i of a handle's text-independent engine (smh_ac.flat_ac chain), NULL past the last: internal, bound for the tests --
 * the CPU suite scans every part with the lane emulator and sums (tests/test_adaptive.py) */
struct smh_ac *smh_ac_flat_part(struct smh_ac *ac, int i);
void smh_wm_dev_free(struct smh_wm_dev *dev); /* smh_runtime.hip */
int smh_wm_prepare_device(struct smh_wm *wm); /* smh_runtime.hip */
int smh_dev_build_peak(int reset);            /* test hook: most table-set builds in flight together */

/* hashes shared by host table build and device lookup -- keep in sync with wm_kernels.hip */
#define SMH_HASH_MUL 0x9E3779B1u
static inline uint32_t smh_fnv1a32(const unsigned char *s, int len)
{
    uint32_t h = 0x811C9DC5u;
    for (int i = 0; i < len; ++i) {
        h ^= s[i];
        h *= 0x01000193u;
    }
    return h;
}

uint32_t smh_wu_shiftsize_for(int alphabet); /* 0 if unsupported */
#define SMH_GRAM_NONE 0
#define SMH_GRAM_PAIR 1
#define SMH_GRAM_BYTE 2
#define SMH_GRAM_OCT 3 /* alphabet 4: 8-symbol grams, table indexed by the gram (16 bits), 8-bit entries, one lookup per column */
/* mixed-length sets on the 4-letter alphabet in ONE pass (pset_host.c): the pair form with TWO plane groups in one
 * 16-bit entry -- group A = the patterns of 14 symbols and more (eight planes of 7-symbol grams over their last 14
 * symbols, 9 entry bits), group B = the shorter ones (J_B = min length - 6 planes over their last J_B + 6 symbols,
 * J_B + 1 entry bits, below A's) -- and two shift-or states per lane, one v_lshl_or each per lookup.  A column is a
 * candidate when either state says so; candidates (rare: few patterns per group, all planes selective) are verified
 * against every length class.  Behind the 128 KiB image: 32 KiB of per-gram values G_A | G_B << 8 for the
 * bounds-checked path. */
#define SMH_GRAM_PAIR2 4
#define SMH_GRAM_OCT2 5
/* 8-bit symbols, round 3: ONE set for the 3-byte grams of all offsets -- a Bloom bit array of 2^20 bits with one hash
 * function (bit = the top 20 bits of gram * SMH_GRAM_MUL mod 2^32, stored INVERTED) -- instead of one plane per offset; the
 * shift-or state then simply asks "were the last J grams all in the set".  J planes of 100 000 keys in 2^17 slots each are
 * 53 % full whatever J is; the flat set holds J x 100 000 keys in 2^20 slots: 25 / 32 / 39 / 45 % full for m = 5 / 6 / 7 / 8
 * (J = m - 2 grams), so 1.6 / 1.0 / 0.9 / 0.8 % of random columns pass instead of 15 / 8 / 4 / 2.2 % -- short byte patterns
 * get a gram filter at all (m = 5..7 had the 12-VALU blocked-Bloom test), m = 8 a better one; from ten grams on the set is
 * fuller than the planes and SMH_GRAM_BYTE wins.  Costs two VALU more per column than SMH_GRAM_BYTE (bit index, bit). */
#define SMH_GRAM_FLAT 6
/* Round 6: SMH_GRAM_BYTE with a table of 143.9 KiB instead of 128 -- what LDS holds beside the 16 KiB of survivor queues the
 * windows-from-L2 verify keeps (wm_kernels.inc smh_gram_lds).  A plane of 100 000 grams is 49 % full instead of 53 %, and eight
 * planes in a row pass 0.35 % of random columns instead of 0.66 %: nearly half the verify stage's work.  The table has no power-of-
 * two size: its index is (v_mul_hi_u32_u24(product, dwords << 8) << 2) | (product >> 30) -- the multiply yields 16 bits, hence
 * dwords, and one v_alignbit puts the product's top two bits under it as the byte -- ONE vector instruction more per column than
 * the top-17-bits index (profiles/r06_final/notes/ab_byte_gram_big_table.log).  Only with the windows-from-L2 verify (no room for
 * staging buffers): sets whose filter passes more than its pipeline takes keep SMH_GRAM_BYTE. */
#define SMH_GRAM_BYTE_BIG 8
/* ... and SMH_GRAM_FLAT in the same 143.9 KiB (1 179 136 bits; read a dword at a time: bit = the product's low five bits, second bit its bits 24..28: the set of 100 000 patterns' grams is 39 % full instead of 43 % at m = 8, six in a row pass 0.34 % instead of 0.63 % */
#define SMH_GRAM_FLAT_BIG 9
/* ... and (late round 6) the one-bit flat set over FOUR-byte grams: three symbols of a 20-letter alphabet are 8000 grams, which a few thousand
 * patterns fill completely (10 000 protein patterns of 8: every 3-gram filter passes 13 % of the columns); four are 160 000.  Product =
 * low three bytes x SMH_GRAM_MUL + fourth byte x SMH_GRAM_MUL4 (one SDWA multiply and one multiply-add), then as SMH_GRAM_FLAT_BIG. */
#define SMH_GRAM_FLAT4_BIG 11
#define SMH_GRAM_MUL4 0x9E3779u
#define SMH_GRAM_PROD4(key32) ((uint32_t)((uint64_t)((key32) & 0xFFFFFFu) * SMH_GRAM_MUL) + (uint32_t)((uint64_t)((uint32_t)(key32) >> 24) * SMH_GRAM_MUL4))
#define SMH_GRAM_BIG_BYTES 147392u
#define SMH_GRAM_BIG_DWORD(prod) (((uint32_t)(((uint64_t)((prod) & 0xFFFFFFu) * (uint64_t)((SMH_GRAM_BIG_BYTES / 4u) << 8)) >> 32)) << 2) /* byte offset of the dword */
#define SMH_GRAM_BIG_INDEX(prod) ((((uint32_t)(((uint64_t)((prod) & 0xFFFFFFu) * (uint64_t)((SMH_GRAM_BIG_BYTES / 4u) << 8)) >> 32)) << 2) | ((uint32_t)(prod) >> 30))
/* candidates per column above which the grouped form is not used.  Round 4: 0.0005 -> 0.002.  A candidate is now decided by the
 * suffix index (one 32-byte record, wm_host.c smh_wm_build_gram_mixed) instead of one window hash, bucket and compare per length
 * class; measured on 1 GiB (profiles/r04_final/notes/mixed_grouped.log): 40 patterns of each length 9..32 (0.0008 per column)
 * 0.231 ms against the joined automaton's 0.291, 10..32 (0.00016) 0.233 / 0.294; at 8..32 (0.0039) the two are level (0.273 /
 * 0.293 in one order of measurement, 0.294 / 0.281 in the other), 100 of each length 8..16 (0.0099) 0.358 against 0.297 */
/* Round 5: 0.002 -> 0.003 with the split of the two groups chosen per set (wm_host.c smh_wm_build_gram_mixed): 40 patterns of
 * each length 8..32 now give 0.0022 candidates per column (split at 11; 0.0039 at the fixed 14) and 0.248 ms against the
 * automaton's 0.286 and the old split's 0.274 (gpurun_out/ab_split.log -> profiles/r05_final/notes/ab_mixed_split.log) */
#define SMH_PSET_GROUPED_DENSITY 0.003
#define SMH_GRAM_PAIR2_SPLIT 14 /* patterns at least this long have all eight planes */
#define SMH_GRAM_BYTES (128u * 1024u)
/* 24-bit multiplier of the byte-gram index (v_mul_u32_u24).  Round 3: 0xD6E8FF instead of the golden-ratio constant
 * 0x9E3779, whose products' MIDDLE bits are badly spread -- over all 2^24 grams, bits 12..31 of gram * 0x9E3779 reach 40 %
 * of their 2^20 values (a random gram then falls on a set bit 2.9 times as often as the set's load says), and bits 15..31
 * load the 2^17 table entries with 67..155 grams each instead of 126..130: 100 000 patterns of 12 bytes let 0.85 % of random
 * columns through the eight planes, with this constant 0.66 % (the ideal for the load: 0.62 %) */
#define SMH_GRAM_MUL 0xD6E8FFu

#ifdef __cplusplus
}
#endif
#endif
