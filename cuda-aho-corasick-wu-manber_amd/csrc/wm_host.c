#define _GNU_SOURCE /* qsort_r */
/*
 * csrc/wm_host.c -- host side of the Wu-Manber path (plain C).
 *
 *   wu_determine_shiftsize / preproc_wu / preproc_wu2
 *        drop-ins for wu/wu.c:18-47, :109-149, :211-251: the caller's SHIFT and
 *        PREFIX_* arrays are filled exactly as the reference fills them.
 *   smh_wm_compile / smh_wm_compile_tables
 *        patterns (+ reference tables) -> device layout of DESIGN.md "WM layout":
 *        an LDS block filter (the device SHIFT table), an HBM verify table (the
 *        device HASH/PREFIX stage) and the reference tables as CSR buckets.
 *
 * How the device layout differs from the reference's (not what it computes):
 *   - PREFIX_value / PREFIX_index are dense [shiftsize x p_size] arrays in the
 *     reference (main.c:436-439: 2 x 2.1 GB at alphabet 256, 100 000 patterns);
 *     here they are packed into CSR buckets before they leave the host;
 *   - the reference's block is fixed at 3 symbols with a 2-bit shift hash
 *     (wu/wu.c:63-67), which on a 4-letter alphabet makes every SHIFT entry 0
 *     from ~1000 patterns up; the device block is as wide as the LDS allows
 *     (up to the whole pattern), so its SHIFT == 0 test actually filters.
 */
#include "smh_internal.h"
#include "hash_engine.h" /* smh_hash_slots */
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

unsigned short m_nBitsInShift = 2; /* smatcher.h:71; main.c:431 sets 2 */
unsigned int shiftsize = 0;        /* smatcher.h:73 */

/* wu/wu.c:18-47: (alphabet-1)*21 + 1 for the alphabets the reference lists */
uint32_t smh_wu_shiftsize_for(int alphabet)
{
    static const int known[] = {2, 4, 8, 20, 128, 256, 512, 1024};
    for (size_t i = 0; i < sizeof known / sizeof known[0]; ++i)
        if (alphabet == known[i]) return (uint32_t)(alphabet - 1) * 21u + 1u;
    return 0;
}

void wu_determine_shiftsize(int alphabet)
{
    uint32_t s = smh_wu_shiftsize_for(alphabet);
    if (!s) fail("The alphabet size is not supported by wu-manber\n");
    shiftsize = s;
}

/* one table build for both pattern representations: stride = m for the flat
 * form, rows[] for the pointer form */
static void wu_fill(const unsigned char *flat, unsigned char *const *rows, int m, int p_size, int B,
                    int nbits, int *SHIFT, int *PREFIX_value, int *PREFIX_index, int *PREFIX_size)
{
    if (B != 3) fail("preproc_wu: the block hash covers 3 symbols, B must be 3\n");
    if (m < 3) fail("preproc_wu: patterns must have at least 3 symbols\n");
    for (int j = 0; j < p_size; ++j) {
        const unsigned char *P = rows ? rows[j] : flat + (size_t)j * m;
        /* blocks ending at offsets m-1 (shift 0) down to B-1 (shift m-B): wu/wu.c:119-131 */
        for (int end = m - 1; end >= B - 1; --end) {
            unsigned h = (((unsigned)P[end - 2] << nbits) + P[end - 1] << nbits) + P[end];
            int shiftlen = m - 1 - end;
            if (shiftlen < SHIFT[h]) SHIFT[h] = shiftlen;
            if (shiftlen == 0) {
                size_t slot = (size_t)h * p_size + (size_t)PREFIX_size[h];
                PREFIX_value[slot] = (int)(((unsigned)P[0] << nbits) + P[1]); /* wu/wu.c:136-138 */
                PREFIX_index[slot] = j;
                PREFIX_size[h]++;
            }
        }
    }
}

void preproc_wu(unsigned char **pattern, int m, int p_size, int alphabet, int B, int *SHIFT,
                int *PREFIX_value, int *PREFIX_index, int *PREFIX_size)
{
    (void)alphabet; /* the reference ignores it too */
    wu_fill(NULL, pattern, m, p_size, B, m_nBitsInShift, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size);
}

void preproc_wu2(unsigned char *pattern_flat, int m, int p_size, int alphabet, int B, int *SHIFT,
                 int *PREFIX_value, int *PREFIX_index, int *PREFIX_size)
{
    (void)alphabet;
    wu_fill(pattern_flat, NULL, m, p_size, B, m_nBitsInShift, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size);
}

/* ------------------------------------------------------------------ device layout */
void smh_wm_host_free(struct smh_wm *wm)
{
    if (!wm) return;
    free(wm->filter);
    free(wm->pair_table);
    free(wm->gram_table);
    free(wm->verify);
    free(wm->verify_ck);
    free(wm->pat_sorted);
    free(wm->l_shift);
    free(wm->l_bucket_off);
    free(wm->l_bucket);
    free(wm->pat_orig);
    smh_ac_free(wm->flex_ac ? wm->flex_ac : wm->alt_ac); /* alt_ac, when set, is the same handle */
    smh_keys_free(wm->keys);
    smh_hash_free(wm->hashes);
    wm->magic = 0;
    free(wm);
}

/* block hash of the hashed filter; the kernels compute the same value with two 24-bit
 * multiplies (smh_wm_block_hash in wm_lane.h) */
static uint32_t smh_wm_block_hash(uint32_t key)
{
    return (uint32_t)((uint64_t)(key & 0xFFFFFFu) * 0x9E3779u) + (uint32_t)((uint64_t)((key >> 8) & 0xFFFFFFu) * 0x85EBCBu);
}

/* hash of a pattern as zero-padded little-endian dwords; the kernels compute it over the text
 * window the same way (smh_wm_mix / smh_window_dword in wm_lane.h) */
static uint32_t smh_wm_tag(const unsigned char *p, int m)
{
    uint32_t h = 0x811C9DC5u;
    for (int j = 0; j < (m + 3) / 4; ++j) {
        uint32_t v = 0;
        for (int b = 0; b < 4 && 4 * j + b < m; ++b) v |= (uint32_t)p[4 * j + b] << (8 * b);
        h = (h ^ v) * 0x9E3779B1u;
        h ^= h >> 15;
    }
    return h;
}

/* row length travels with the call (qsort_r): concurrent compiles with different m share nothing */
static int cmp_rows(const void *a, const void *b, void *m) { return memcmp(a, b, (size_t)*(const int *)m); }

static int ceil_log2_u32(uint32_t v)
{
    int b = 0;
    while ((1ull << b) < v) ++b;
    return b;
}

/* code of the `w` symbols ending at s[0], oldest symbol in the highest bits --
 * the value the kernels hold in their rolling register (wm_kernels.hip) */
static uint32_t block_code(const unsigned char *s_last, int w, int bits)
{
    uint64_t code = 0;
    for (int i = w - 1; i >= 0; --i) code = (code << bits) | s_last[-i];
    return (uint32_t)code;
}

#define SMH_WM_FILTER_LOG2_MAX 20 /* 2^20 bits = 128 KiB of LDS */

/* ---- gram filter (smh_internal.h): plane j holds the q-grams that end j symbols before the patterns' ends.
 * Planes of overlapping grams do not pass independently (two grams that share q-1 symbols are both in their
 * planes far more often than the product of the plane loads says: 8000 DNA patterns, 7-symbol grams, eight
 * planes 39 % full each: 0.41 % of random columns survive, not 0.05 %), so the fraction of columns that will
 * reach the verify stage is MEASURED here by running the recurrence over pseudo-random text. */
static uint64_t gram_rng(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define SMH_GRAM_FLAT_K2 (-6) /* gram_survivors only: the flat form with two bits per gram */
#define SMH_GRAM_FLAT_BIG_K2 (-9) /* ... in the 143.9 KiB table */
static double gram_survivors(int kind, const void *tab, int alphabet, int planes)
{
    enum { COLS = 1 << 18 };
    uint64_t seed = 0x5EEDull, hits = 0;
    uint32_t S = ~0u, code = 0, k0 = 0, k1 = 0, k2s = 0;
    const int cand_bit = kind == SMH_GRAM_PAIR ? planes - 1 : 7; /* the pair form holds J-bit values, candidate = bit J-1 */
    if (kind == SMH_GRAM_OCT2) { /* a lookup every second column, two END columns decided per lookup */
        for (int x = 0; x < COLS; ++x) {
            code = ((code << 2) | (uint32_t)(gram_rng(&seed) & 3u)) & 0xFFFFu;
            if (!(x & 1)) continue;
            S = (S << 2) | ((const uint16_t *)tab)[code];
            if (x >= 64) hits += (((S >> (planes - 1)) & 1u) ^ 1u) + (((S >> (planes - 2)) & 1u) ^ 1u);
        }
        return (double)hits / (double)(COLS - 64);
    }
    for (int x = 0; x < COLS; ++x) {
        const uint32_t c = (uint32_t)(gram_rng(&seed) % (uint64_t)alphabet);
        uint32_t G;
        if (kind == SMH_GRAM_PAIR) {
            code = ((code << 2) | c) & 0x3FFFu;                     /* the 7 symbols ending here */
            G = ((const uint16_t *)((const uint8_t *)tab + SMH_GRAM_BYTES))[code]; /* the per-gram values behind the LDS image */
        } else if (kind == SMH_GRAM_OCT) {
            code = ((code << 2) | c) & 0xFFFFu;
            G = ((const uint8_t *)tab)[code];
        } else if (kind == SMH_GRAM_FLAT4_BIG) { /* the flat set over four-byte grams (a dword of the set, bit = the product's low five bits) */
            const uint32_t key = k0 | (k1 << 8) | (k2s << 16) | (c << 24);
            k0 = k1;
            k1 = k2s;
            k2s = c;
            const uint32_t prod = SMH_GRAM_PROD4(key);
            uint32_t word;
            memcpy(&word, (const uint8_t *)tab + SMH_GRAM_BIG_DWORD(prod), 4);
            G = ((word >> (prod & 31u)) & 1u) ? (0xFFu & ~((1u << (8 - planes)) - 1u)) : 0u;
        } else if (kind == SMH_GRAM_FLAT || kind == SMH_GRAM_FLAT_K2 || kind == SMH_GRAM_FLAT_BIG || kind == SMH_GRAM_FLAT_BIG_K2) {
            const uint32_t key = k0 | (k1 << 8) | (c << 16);
            k0 = k1;
            k1 = c;
            const int fbig = kind == SMH_GRAM_FLAT_BIG || kind == SMH_GRAM_FLAT_BIG_K2, fk2 = kind == SMH_GRAM_FLAT_K2 || kind == SMH_GRAM_FLAT_BIG_K2;
            const uint32_t prod = (uint32_t)((uint64_t)key * SMH_GRAM_MUL);
            uint32_t out;
            if (fbig) { /* a dword of the set, bit = the product's low five bits (wm_lane.h smh_flat_group) */
                uint32_t word;
                memcpy(&word, (const uint8_t *)tab + SMH_GRAM_BIG_DWORD(prod), 4);
                out = (word >> (prod & 31u)) | (fk2 ? word >> ((prod >> 24) & 31u) : 0u);
            } else {
                const uint32_t b = ((const uint8_t *)tab)[prod >> 15];
                out = (b >> ((prod >> 12) & 7u)) | (fk2 ? b >> ((prod >> 9) & 7u) : 0u);
            }
            G = (out & 1u) ? (0xFFu & ~((1u << (8 - planes)) - 1u)) : 0u;
        } else {
            const uint32_t key = k0 | (k1 << 8) | (c << 16);
            k0 = k1;
            k1 = c;
            const uint32_t prod = (uint32_t)((uint64_t)key * SMH_GRAM_MUL);
            G = ((const uint8_t *)tab)[kind == SMH_GRAM_BYTE_BIG ? SMH_GRAM_BIG_INDEX(prod) : prod >> 15];
        }
        S = (S << 1) | G;                                           /* shift-or: candidate bit clear = candidate */
        if (x >= 32) hits += ((S >> cand_bit) & 1u) ^ 1u;
    }
    return (double)hits / (double)(COLS - 32);
}

/* scan time (ms per GiB on MI355X, profiles/r02_*) of each form + what a surviving column costs in the verify
 * stage; the form with the lowest estimate is kept, or none when the handle's block filter is estimated faster.
 * Shift-or steps (one v_lshl_or per column, per two columns in the pair form): pairs 0.178 (HBM-bound), 8-symbol
 * grams 0.26 and byte grams 0.238 (one LDS lookup per column). */
#define SMH_GRAM_PAIR_MS 0.173
#define SMH_GRAM_OCT_MS 0.24 /* 0.218 at 2000 patterns .. 0.243 at 12 000 with next to no survivors (lane 0 of a wave-chunk keeps the assumption) */
#define SMH_GRAM_OCT2_MS 0.178 /* the pair form's lookups + lane 0's inherited state from the halo in every chunk */
#define SMH_GRAM_BYTE_MS 0.238
#define SMH_GRAM_BYTE_BIG_MS 0.243 /* one vector instruction more per column (the index), under the LDS lookup's shadow */
#define SMH_GRAM_FLAT_BIG_MS 0.259 /* the same instruction count as SMH_GRAM_FLAT (the set is read a dword at a time); with the survivors dropped it runs
                                    * 0.92 ms per 4 GiB where SMH_GRAM_BYTE_BIG runs 0.855: +0.016 ms/GiB.  (0.272 until late in round 6: m = 9 -- seven
                                    * grams, 15 survivors per chunk against the hashed planes' 28 -- went to the planes by 0.005 and ran 9 % slower) */
#define SMH_GRAM_FLAT_MS 0.27 /* SMH_GRAM_BYTE's lookup per column + two VALU (bit index, bit) */
#define SMH_GRAM_FLAT4_BIG_MS 0.272 /* SMH_GRAM_FLAT_BIG + one vector instruction per column (the fourth byte's multiply; the add rides on the first) */
/* verify stage, ms per GiB for a fraction `dens` of surviving columns.  Staged (m <= 33: window hashes from the LDS
 * copy of the chunk, probe pipelined): the cost is mostly per wave-chunk that has any survivor -- lock, copy, hash
 * round trips -- and grows slowly with their number (pair form, 16 symbols, survivors per 4 KiB chunk -> ms/GiB over the
 * bare scan: 0.08 -> 0.002, 0.25 -> 0.02, 1.1 -> 0.05, 4.4 -> 0.09, 17 -> 0.13; byte grams, 100 000 patterns: 35 ->
 * 0.11 / 0.15 at 12 / 20 symbols, 116 -> 0.28, 690 -> 1.28; gpurun_out/r02_s, r02_z .. r02_ac).  Windows re-read from HBM (longer patterns): linear in
 * the survivors and the window the stage has to fetch and hash (fits to 8000 DNA patterns / 100 000 byte patterns
 * before staging: 0.49 / 0.72, 0.63 / 0.82). */
/* the flat set in the big table: clear bit `bit` (0..31) of the little-endian dword the product selects */
static void flat_big_clear(uint8_t *tab, uint32_t prod, uint32_t bit)
{
    tab[SMH_GRAM_BIG_DWORD(prod) + (bit >> 3)] &= (uint8_t)~(1u << (bit & 7u));
}
static double gram_verify_ms(int m, double dens)
{
    if (m > 33) return (12.0 + 3.0 * m) * dens;
    const double pc = dens * 4096.0; /* survivors per wave-chunk; beyond a queue's worth several flushes per chunk */
    return 0.04 * log(1.0 + 1.5 * pc) + (pc > 30.0 ? 0.0015 * (pc - 30.0) : 0.0);
}
/* the same for the pair-like forms (two columns per lookup), refit at steady state on 35 DNA sets x 3 forms (2000..40 000
 * patterns of 12..28 symbols, profiles/r03_experiments/gram_forms_sweep.log): up to SMH_REGV_MAX_PER_CHUNK survivors per chunk
 * they verify in registers, 0.02 / 0.045 / 0.07 ms/GiB at 1.5 / 4.4 / 8 per chunk -- 0.7 of the staged model -- and between
 * 8 and 40 per chunk the staged verify measures 1.2 of it (0.137 at 8.8, m = 12).  With these and SMH_GRAM_OCT_MS the form
 * the compile keeps is the fastest of the three, or within 1.5 % of it, on 33 of the 35 sets (the other two: 4.5 %) */
static double gram_verify_ms_pairlike(int m, double dens)
{
    const double pc = dens * 4096.0;
    /* late round 6: between SMH_L2_MIN_PER_CHUNK and 40 survivors per chunk these forms verify through the windows-from-L2 pipeline
     * (wm_kernels.inc launch_gram), fit on 13 DNA sets x 3 forms pinned to the filter kernels (tools/l2_fit.py,
     * profiles/r06_final/notes/ab_dna_l2_verify.log): 0.014 / 0.037 / 0.079 / 0.082 / 0.158 ms/GiB over the bare scan at 1.2 / 6.2 /
     * 13.8 / 19.9 / 34 per chunk */
    if (m <= 33 && pc >= SMH_L2_MIN_PER_CHUNK_REGV && pc <= SMH_L2_DNA_MAX_PER_CHUNK) return 0.01 + 0.0045 * pc;
    const double f = m > 33 ? 1.0 : pc <= SMH_REGV_MAX_PER_CHUNK ? 0.7 : pc <= 40.0 ? 1.2 : 1.0;
    return f * gram_verify_ms(m, dens);
}
/* the 8-symbol-gram form (one lookup per column) with that pipeline: 0.029 / 0.033 / 0.060 / 0.125 over its bare 0.215 at 0.25 / 1.7 /
 * 5.9 / 36 per chunk -- over SMH_GRAM_OCT_MS, which holds 0.025 of it */
static double gram_verify_ms_oct(int m, double dens)
{
    const double pc = dens * 4096.0;
    if (m <= 33 && pc >= SMH_L2_MIN_PER_CHUNK && pc <= SMH_L2_DNA_MAX_PER_CHUNK) return 0.005 + 0.0027 * pc;
    return gram_verify_ms(m, dens);
}
#define SMH_HASHED_VERIFY_MS(m) (6.0 + 1.05 * (m))
#define SMH_DIRECT_VERIFY_MS(m) ((m) > 4 ? 1.9 * (m) - 3.0 : 4.6)

static int build_gram_filter(struct smh_wm *wm, double other_ms)
{
    const int m = wm->m, d = wm->distinct;
    const unsigned char *pats = wm->pat_sorted;
    void *best = NULL;
    int best_kind = SMH_GRAM_NONE, best_planes = 0;
    uint32_t best_bytes = 0;
    double best_ms = other_ms, best_dens = 0.0;
    /* development knob: SMH_WM_TUNE="gram=K" keeps form K whenever the set can use it (0: never a gram filter) */
    const int force = smh_tune_int(SMH_TUNE_WM, "gram=", -1);
    if (force == 0) return 0;
    if (force > 0) best_ms = 1e30;
#define GRAM_WANTED(kind) (force < 0 || force == (kind))
    if (wm->alphabet == 4 && m >= 9 && GRAM_WANTED(SMH_GRAM_PAIR)) {
        /* 7-symbol grams; planes 0 .. J-1 need the gram that ends j before the end to start at >= 0.  A 16-bit entry
         * holds (G of the older seven << 1) | G of the newer seven, so up to FIFTEEN planes: with J = m - 6 the chain of
         * overlapping grams covers the whole pattern and next to nothing but matches survives (8000 patterns of 16
         * symbols: eight planes let 0.41 % of random columns through, ten planes 0.0002 %) */
        int J = m - 6;
        if (J > 15) J = 15;
        /* LDS image (128 KiB of 16-bit entries) + the per-gram values G (32 KiB, HBM only: bounds-checked path) */
        uint16_t *tab = (uint16_t *)malloc(SMH_GRAM_BYTES + 32768);
        if (!tab) return -1;
        uint16_t *g7 = (uint16_t *)((uint8_t *)tab + SMH_GRAM_BYTES);
        for (uint32_t c = 0; c < 16384; ++c) g7[c] = (uint16_t)((1u << J) - 1u); /* bit J-1-j SET = the gram is NOT in plane j */
        for (int p = 0; p < d; ++p)
            for (int j = 0; j < J; ++j) {
                const unsigned char *g = pats + (size_t)p * m + (m - 7 - j);
                uint32_t code = 0;
                for (int i = 0; i < 7; ++i) code = (code << 2) | g[i];
                g7[code] &= (uint16_t)~(1u << (J - 1 - j));
            }
        /* entry of eight symbols = (G of the older seven << 1) | G of the newer seven: one v_lshl_or does both columns */
        for (uint32_t x = 0; x < 65536; ++x) tab[x] = (uint16_t)(((uint32_t)g7[x >> 2] << 1) | g7[x & 0x3FFFu]);
        const double dens = gram_survivors(SMH_GRAM_PAIR, tab, 4, J), ms = SMH_GRAM_PAIR_MS + gram_verify_ms_pairlike(m, dens);
        /* lane 0 of a wave-chunk starts from "every plane still alive": its column c passes on planes 0..c alone */
        double lane0 = 0.0, run = 1.0;
        for (int j = 0; j < J - 1; ++j) {
            uint32_t in = 0;
            for (uint32_t c = 0; c < 16384; ++c) in += !((g7[c] >> (J - 1 - j)) & 1u);
            run *= (double)in / 16384.0;
            lane0 += run;
        }
        if (ms < best_ms) {
            free(best);
            best = tab; best_kind = SMH_GRAM_PAIR; best_planes = J; best_bytes = SMH_GRAM_BYTES + 32768; best_ms = ms; best_dens = dens;
            wm->gram_lane0 = lane0;
        } else {
            free(tab);
        }
    }
    if (wm->alphabet == 4 && m >= 10 && GRAM_WANTED(SMH_GRAM_OCT)) {
        int J = m - 7;
        if (J > 8) J = 8;
        uint8_t *tab = (uint8_t *)malloc(65536);
        if (!tab) { free(best); return -1; }
        memset(tab, 0xFF & ~((1 << (8 - J)) - 1), 65536);
        for (int p = 0; p < d; ++p)
            for (int j = 0; j < J; ++j) {
                const unsigned char *g = pats + (size_t)p * m + (m - 8 - j);
                uint32_t code = 0;
                for (int i = 0; i < 8; ++i) code = (code << 2) | g[i];
                tab[code] &= (uint8_t)~(1u << (7 - j));
            }
        const double dens = gram_survivors(SMH_GRAM_OCT, tab, 4, J), ms = SMH_GRAM_OCT_MS + gram_verify_ms_oct(m, dens);
        if (ms < best_ms) {
            free(best);
            best = tab; best_kind = SMH_GRAM_OCT; best_planes = J; best_bytes = 65536; best_ms = ms; best_dens = dens;
        } else {
            free(tab);
        }
    }
    if (wm->alphabet == 4 && m >= 10 && m <= 33 && GRAM_WANTED(SMH_GRAM_OCT2)) {
        /* 8-symbol grams, a lookup per two columns (smh_internal.h SMH_GRAM_OCT2); m <= 33: lane 0 of a wave-chunk needs the
         * state it inherits in EVERY chunk (its first column's candidate bit comes from the lookup in front of the chunk),
         * which the kernels work out from the 16 or 32 bytes of halo they keep */
        int J = m - 7;
        if (J > 16) J = 16;
        uint16_t *tab = (uint16_t *)malloc(SMH_GRAM_BYTES);
        if (!tab) { free(best); return -1; }
        for (uint32_t c = 0; c < 65536; ++c) tab[c] = (uint16_t)((1u << J) - 1u); /* bit J-1-j SET = the gram is NOT in plane j */
        for (int p = 0; p < d; ++p)
            for (int j = 0; j < J; ++j) {
                const unsigned char *g = pats + (size_t)p * m + (m - 8 - j);
                uint32_t code = 0;
                for (int i = 0; i < 8; ++i) code = (code << 2) | g[i];
                tab[code] &= (uint16_t)~(1u << (J - 1 - j));
            }
        const double dens = gram_survivors(SMH_GRAM_OCT2, tab, 4, J), ms = SMH_GRAM_OCT2_MS + gram_verify_ms_pairlike(m, dens);
        if (ms < best_ms) {
            free(best);
            best = tab; best_kind = SMH_GRAM_OCT2; best_planes = J; best_bytes = SMH_GRAM_BYTES; best_ms = ms; best_dens = dens;
        } else {
            free(tab);
        }
    }
    /* (round 4: from nine symbols up, not only for byte alphabets -- a symbol is a byte of text whatever the alphabet, and on
     * the 20-letter alphabet a plane of 1000 patterns' 3-symbol grams holds 12 % of the 8000 possible ones: eight planes let
     * nothing through, where the direct filter on the last four symbols passed 0.6 % of the columns to the verify stage) */
    for (int big = 0; big <= 1; ++big) {
        /* SMH_GRAM_BYTE, and (round 6) the same planes in the 143.9 KiB table SMH_GRAM_BYTE_BIG: windows of up to 33 bytes (its verify
         * is the windows-from-L2 pipeline) and sets that leave its pipeline room (launch_gram, wm_kernels.inc: 56 per 4 KiB chunk) */
        const int kind = big ? SMH_GRAM_BYTE_BIG : SMH_GRAM_BYTE;
        if (!(wm->bits_per_symbol >= 4 && m >= 5 && GRAM_WANTED(kind))) continue;
        if (big && m - 1 > 32) continue;
        const size_t bytes = big ? SMH_GRAM_BIG_BYTES : SMH_GRAM_BYTES;
        int J = m - 2;
        if (J > 8) J = 8;
        uint8_t *tab = (uint8_t *)malloc(bytes);
        if (!tab) { free(best); return -1; }
        memset(tab, 0xFF & ~((1 << (8 - J)) - 1), bytes);
        for (int p = 0; p < d; ++p)
            for (int j = 0; j < J; ++j) {
                const unsigned char *g = pats + (size_t)p * m + (m - 3 - j);
                const uint32_t key = (uint32_t)g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16);
                const uint32_t prod = (uint32_t)((uint64_t)key * SMH_GRAM_MUL); /* low 32 bits of the product */
                const uint32_t idx = big ? SMH_GRAM_BIG_INDEX(prod) : prod >> 15; /* (the top 17) */
                tab[idx] &= (uint8_t)~(1u << (7 - j));
            }
        const double dens = gram_survivors(kind, tab, wm->alphabet, J), ms = (big ? SMH_GRAM_BYTE_BIG_MS : SMH_GRAM_BYTE_MS) + gram_verify_ms(m, dens);
        if (big && force != SMH_GRAM_BYTE_BIG && dens * 4096.0 > 56.0) { free(tab); continue; } /* (its pipeline would overflow -- launch_gram, wm_kernels.inc: 56 per chunk -- and the staged verify of SMH_GRAM_BYTE is the faster one there) */
        if (ms < best_ms) {
            free(best);
            best = tab; best_kind = kind; best_planes = J; best_bytes = (uint32_t)bytes; best_ms = ms; best_dens = dens;
        } else {
            free(tab);
        }
    }
    for (int big = 0; big <= 1; ++big) {
        /* one Bloom set for the grams of all offsets (smh_internal.h SMH_GRAM_FLAT); bit = 1: NOT in the set.  Round 6: also in the
         * 143.9 KiB table (SMH_GRAM_FLAT_BIG: windows of up to 33 bytes, sets that leave the L2 pipeline room, as SMH_GRAM_BYTE_BIG) */
        const int kind = big ? SMH_GRAM_FLAT_BIG : SMH_GRAM_FLAT;
        if (!(wm->bits_per_symbol >= 4 && m >= 5 && GRAM_WANTED(kind))) continue;
        if (big && m - 1 > 32) continue;
        const size_t bytes = big ? SMH_GRAM_BIG_BYTES : SMH_GRAM_BYTES;
        int J = m - 2;
        if (J > 8) J = 8;
        uint8_t *tab = (uint8_t *)malloc(bytes);
        if (!tab) { free(best); return -1; }
        memset(tab, 0xFF, bytes);
        for (int p = 0; p < d; ++p)
            for (int j = 0; j < J; ++j) {
                const unsigned char *g = pats + (size_t)p * m + (m - 3 - j);
                const uint32_t key = (uint32_t)g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16);
                const uint32_t prod = (uint32_t)((uint64_t)key * SMH_GRAM_MUL);
                if (big) flat_big_clear(tab, prod, prod & 31u);
                else tab[prod >> 15] &= (uint8_t)~(1u << ((prod >> 12) & 7u));
            }
        if (smh_tune_has(SMH_TUNE_WM, "debug")) {
            uint64_t zeros = 0;
            for (uint32_t i = 0; i < bytes; ++i) zeros += 8u - (uint32_t)__builtin_popcount(tab[i]);
            fprintf(stderr, "flat byte grams: %d patterns x %d grams, %.1f %% of the %zu bits in the set\n", d, J, 100.0 * (double)zeros / (8.0 * (double)bytes), 8 * bytes);
        }
        const double flat_ms = big ? SMH_GRAM_FLAT_BIG_MS : SMH_GRAM_FLAT_MS;
        double dens = gram_survivors(big ? SMH_GRAM_FLAT_BIG : SMH_GRAM_FLAT, tab, wm->alphabet, J), ms = flat_ms + gram_verify_ms(m, dens);
        int k2 = 0;
        if (J <= 5) {
            /* round 4: two bits per gram in its byte (wm_lane.h smh_flat_group<.., K2>) while the grams are few enough for the fuller
             * array to pay -- 100 000 patterns of 5 bytes: 1.6 % -> 0.7 % of random columns pass; kept when it measures better */
            uint8_t *t2 = (uint8_t *)malloc(bytes);
            if (t2) {
                memset(t2, 0xFF, bytes);
                for (int p = 0; p < d; ++p)
                    for (int j = 0; j < J; ++j) {
                        const unsigned char *g = pats + (size_t)p * m + (m - 3 - j);
                        const uint32_t key = (uint32_t)g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16);
                        const uint32_t prod = (uint32_t)((uint64_t)key * SMH_GRAM_MUL);
                        if (big) { flat_big_clear(t2, prod, prod & 31u); flat_big_clear(t2, prod, (prod >> 24) & 31u); }
                        else t2[prod >> 15] &= (uint8_t)~((1u << ((prod >> 12) & 7u)) | (1u << ((prod >> 9) & 7u)));
                    }
                /* what the second bit costs the scan: 0.01 ms/GiB in the byte-addressed form (round 4), 0.022 in the big table (filter
                 * alone 0.894 -> 0.953 ms per 4 GiB).  The big form's windows-from-L2 verify costs about the same at 16 and at 28
                 * survivors per chunk (0.05 / 0.057 ms/GiB) and twice that at 46, where its queue overflows: there the second bit
                 * pays (100 000 patterns, m = 5: 46 -> 26 per chunk, 1.44 -> 1.19 ms per 4 GiB) and below it loses (m = 6: 28 -> 16,
                 * 1.15 -> 1.21; m = 7: 1.09 -> 1.23) -- profiles/r06_final/notes/ab_byte_gram_big_table.log */
                const double d2 = gram_survivors(big ? SMH_GRAM_FLAT_BIG_K2 : SMH_GRAM_FLAT_K2, t2, wm->alphabet, J), ms2 = flat_ms + (big ? 0.022 : 0.01) + gram_verify_ms(m, d2);
                const int fk = smh_tune_int(SMH_TUNE_WM, "flatk=", 0); /* development knob "flatk=1|2": one / two bits per gram regardless */
                if (smh_tune_has(SMH_TUNE_WM, "debug")) fprintf(stderr, "flat byte grams%s: one bit per gram %.5f of the columns survive, est %.3f ms/GiB; two bits %.5f, est %.3f\n", big ? " (big table)" : "", dens, ms, d2, ms2);
                if (fk == 2 || (fk != 1 && ms2 < ms && (!big || dens * 4096.0 > 30.0))) {
                    free(tab);
                    tab = t2; dens = d2; ms = ms2; k2 = 1;
                } else {
                    free(t2);
                }
            }
        }
        if (big && force != SMH_GRAM_FLAT_BIG && dens * 4096.0 > 56.0) { free(tab); continue; } /* (the L2 pipeline would overflow: SMH_GRAM_FLAT with its staged verify) */
        if (ms < best_ms) {
            free(best);
            best = tab; best_kind = kind; best_planes = J; best_bytes = (uint32_t)bytes; best_ms = ms; best_dens = dens;
            wm->gram_jb = k2;
        } else {
            free(tab);
        }
    }
    /* late round 6: the flat set over FOUR-byte grams (smh_internal.h SMH_GRAM_FLAT4_BIG), one bit per gram, in the 143.9 KiB table with its
     * windows-from-L2 verify: for alphabets whose three-symbol grams the set saturates.  J = m - 3 grams in a row. */
    if (wm->bits_per_symbol >= 4 && m >= 6 && m - 1 <= 32 && GRAM_WANTED(SMH_GRAM_FLAT4_BIG)) {
        int J = m - 3;
        if (J > 8) J = 8;
        uint8_t *tab = (uint8_t *)malloc(SMH_GRAM_BIG_BYTES);
        if (!tab) { free(best); return -1; }
        memset(tab, 0xFF, SMH_GRAM_BIG_BYTES);
        for (int p = 0; p < d; ++p)
            for (int j = 0; j < J; ++j) {
                const unsigned char *g = pats + (size_t)p * m + (m - 4 - j);
                const uint32_t key = (uint32_t)g[0] | ((uint32_t)g[1] << 8) | ((uint32_t)g[2] << 16) | ((uint32_t)g[3] << 24);
                const uint32_t prod = SMH_GRAM_PROD4(key);
                flat_big_clear(tab, prod, prod & 31u);
            }
        const double dens = gram_survivors(SMH_GRAM_FLAT4_BIG, tab, wm->alphabet, J), ms = SMH_GRAM_FLAT4_BIG_MS + gram_verify_ms(m, dens);
        if (smh_tune_has(SMH_TUNE_WM, "debug")) fprintf(stderr, "flat four-byte grams (big table): %.5f of the columns survive, est %.3f ms/GiB\n", dens, ms);
        if ((force == SMH_GRAM_FLAT4_BIG || dens * 4096.0 <= 56.0) && ms < best_ms) {
            free(best);
            best = tab; best_kind = SMH_GRAM_FLAT4_BIG; best_planes = J; best_bytes = SMH_GRAM_BIG_BYTES; best_ms = ms; best_dens = dens;
            wm->gram_jb = 0;
        } else {
            free(tab);
        }
    }
#undef GRAM_WANTED
    if (smh_tune_has(SMH_TUNE_WM, "debug"))
        fprintf(stderr, "gram filter: block filter est %.3f ms/GiB; kept form %d, %d planes, survivors %.5f, est %.3f ms/GiB\n",
                other_ms, best_kind, best_planes, best_dens, best_ms);
    wm->gram_kind = best_kind;
    wm->gram_planes = best_planes;
    wm->gram_table = best;
    wm->gram_bytes = best_bytes;
    wm->gram_density = best_dens;
    wm->scan_ms_est = best_ms;
    return 0;
}

/* the two groups' per-gram plane bytes for one way of splitting the set -- group A = the patterns of `split` symbols and more
 * with JA = min(8, shortest of them - 6) planes (the deeper planes accept every gram, so the eight-plane recurrence of the lane code
 * decides on JA), group B = the shorter ones with JB = min(5, shortest - 6) -- and the candidates per column on pseudo-random text */
static double grouped_planes(const unsigned char *patterns, const uint32_t *lengths, int p_size, uint32_t split, uint8_t *gA, uint8_t *gB, int *JB_out)
{
    uint32_t minA = UINT32_MAX, minB = UINT32_MAX;
    for (int p = 0; p < p_size; ++p) {
        if (lengths[p] >= split) { if (lengths[p] < minA) minA = lengths[p]; }
        else if (lengths[p] < minB) minB = lengths[p];
    }
    int JA = minA == UINT32_MAX ? 8 : (int)minA - 6, JB = minB == UINT32_MAX ? 0 : (int)minB - 6;
    if (JA > 8) JA = 8;
    if (JB > 5) JB = 5;
    memset(gA, minA == UINT32_MAX ? 0xFF : 0xFF & ~((1 << (8 - JA)) - 1), 16384); /* bit 7-j SET = the gram is not in plane j of group A */
    memset(gB, JB ? (1 << JB) - 1 : 0, 16384);                                   /* bit JB-1-j SET = not in plane j of group B */
    uint64_t off = 0;
    for (int p = 0; p < p_size; ++p) {
        const uint32_t L = lengths[p];
        const unsigned char *pat = patterns + off;
        off += L;
        const int isA = L >= split, J = isA ? JA : JB;
        for (int j = 0; j < J; ++j) {
            const unsigned char *g = pat + (L - 7 - (uint32_t)j);
            uint32_t code = 0;
            for (int i = 0; i < 7; ++i) code = (code << 2) | g[i];
            if (isA) gA[code] &= (uint8_t)~(1u << (7 - j));
            else gB[code] &= (uint8_t)~(1u << (JB - 1 - j));
        }
    }
    enum { COLS = 1 << 18 };
    uint64_t seed = 0x5EEDull, hits = 0;
    uint32_t SA = ~0u, SB = ~0u, code = 0;
    for (int x = 0; x < COLS; ++x) {
        code = ((code << 2) | (uint32_t)(gram_rng(&seed) & 3u)) & 0x3FFFu;
        SA = (SA << 1) | gA[code];
        SB = (SB << 1) | gB[code];
        if (x >= 16) hits += (((SA >> 7) & 1u) ^ 1u) | (JB ? ((SB >> (JB - 1)) & 1u) ^ 1u : 0u);
    }
    *JB_out = JB;
    return (double)hits / (double)(COLS - 16);
}

int smh_wm_build_gram_mixed(struct smh_wm *suffix, const unsigned char *patterns, const uint32_t *lengths, int p_size)
{
    if (!suffix || suffix->alphabet != 4 || p_size < 1) return 1;
    uint32_t minlen = UINT32_MAX;
    for (int p = 0; p < p_size; ++p)
        if (lengths[p] < minlen) minlen = lengths[p];
    if (minlen < 8) return 1; /* a pattern of 7 symbols has one plane: nothing to chain */
    uint16_t *tab = (uint16_t *)malloc(SMH_GRAM_BYTES + 32768u);
    uint8_t *gA = (uint8_t *)malloc(16384), *gB = (uint8_t *)malloc(16384);
    if (!tab || !gA || !gB) { free(tab); free(gA); free(gB); return -1; }
    /* Round 5: where the set is split is chosen, not fixed at SMH_GRAM_PAIR2_SPLIT.  Group B's depth is its SHORTEST pattern's, so
     * with lengths 8..32 a split at 14 tests the 200 patterns of 9..13 symbols on their last eight only (0.0039 candidates per
     * column); split at 10 or 11 they join group A with 4 or 5 planes and the 80 or 120 shortest are left to B: 0.0020. */
    uint32_t split = SMH_GRAM_PAIR2_SPLIT;
    int JB = 0;
    double dens = 2.0;
    for (uint32_t sp = 9; sp <= SMH_GRAM_PAIR2_SPLIT; ++sp) {
        if (sp != SMH_GRAM_PAIR2_SPLIT && sp <= minlen) continue; /* (no pattern below it: the same as any other such split) */
        int jb;
        const double d = grouped_planes(patterns, lengths, p_size, sp, gA, gB, &jb);
        if (d < dens) { dens = d; split = sp; }
    }
    if (smh_tune_has(SMH_TUNE_WM, "split14")) split = SMH_GRAM_PAIR2_SPLIT; /* development knob: the round-4 split */
    dens = grouped_planes(patterns, lengths, p_size, split, gA, gB, &JB);
    int short_ones = 0;
    for (int p = 0; p < p_size; ++p) short_ones += lengths[p] < split;
    const int bsh = JB ? JB + 1 : 0;
    for (uint32_t x = 0; x < 65536; ++x) {
        const uint32_t eA = ((uint32_t)gA[x >> 2] << 1) | gA[x & 0x3FFFu], eB = ((uint32_t)gB[x >> 2] << 1) | gB[x & 0x3FFFu];
        tab[x] = (uint16_t)((eA << bsh) | eB);
    }
    uint16_t *gx = (uint16_t *)((uint8_t *)tab + SMH_GRAM_BYTES);
    for (uint32_t c = 0; c < 16384; ++c) gx[c] = (uint16_t)(gA[c] | ((uint32_t)gB[c] << 8));
    free(gA); free(gB);
    if (smh_tune_has(SMH_TUNE_WM, "debug"))
        fprintf(stderr, "grouped pair-gram filter: split at %u, %d short patterns with %d planes, candidates %.6f per column\n", split, short_ones, JB, dens);
    /* every candidate is looked up in the suffix index (window from HBM, one record), and a group's planes together are
     * as selective as an exact match of its shortest pattern's length: with many SHORT patterns the candidates are
     * mostly real matches of those classes and an automaton counts them in line, cheaper (pset_host.c falls back) */
    if (dens > SMH_PSET_GROUPED_DENSITY && !smh_tune_has(SMH_TUNE_WM, "grouped=force")) { free(tab); return 1; }
    /* The verify stage's index (round 4): every pattern keyed by its LAST EIGHT symbols -- what every candidate column of
     * either group has matched at the least.  One 32-byte record per 8-symbol code: {next record + 1, length, where the whole
     * pattern lies, 0, the pattern's last 16 bytes END-aligned}; patterns that share their last eight symbols chain through
     * overflow records of the same layout; the patterns themselves END-aligned in whole dwords behind them.  A candidate costs
     * one record -- which decides every pattern of up to 16 symbols by itself -- however many length classes the set has;
     * verified class by class (a window hash, a bucket and a compare per class, one after the other) the headline's 25
     * classes at 15.8 candidates per 4 KiB took 2.9 ms/GiB.  Identical patterns are entered once: the count is the number of
     * DISTINCT patterns ending at a column, as the classes' is. */
    {
        uint64_t pat_dw = 0;
        for (int p = 0; p < p_size; ++p) pat_dw += (lengths[p] + 3u) >> 2;
        const size_t base = SMH_GRAM_BYTES + 32768u, slot_bytes = 65536u * 32u, ent_bytes = (size_t)p_size * 32u;
        const size_t total = base + slot_bytes + ent_bytes + (size_t)pat_dw * 4u;
        if (total > 0x7FFFFFFFu) { free(tab); return 1; }
        uint16_t *grown = (uint16_t *)realloc(tab, total);
        if (!grown) { free(tab); return -1; }
        tab = grown;
        uint32_t *slot = (uint32_t *)((uint8_t *)tab + base), *ent = slot + 65536u * 8u, *pw = ent + (size_t)p_size * 8u;
        memset(slot, 0, total - base);
        uint32_t n_ent = 0, dw = 0;
        uint64_t off = 0;
        for (int p = 0; p < p_size; ++p) {
            const uint32_t L = lengths[p], nd = (L + 3u) >> 2;
            const unsigned char *pat = patterns + off;
            off += L;
            uint32_t code = 0;
            for (uint32_t i = L - 8u; i < L; ++i) code = (code << 2) | (pat[i] & 3u);
            uint8_t *full = (uint8_t *)(pw + dw); /* the pattern's last byte in the record's last byte */
            memcpy(full + (4u * nd - L), pat, L);
            int dup = 0;
            for (uint32_t *r = slot + 8u * code; r && r[1] && !dup; r = r[0] ? ent + 8u * (r[0] - 1u) : NULL)
                dup = r[1] == L && !memcmp(pw + r[2], full, 4u * nd);
            if (dup) { memset(full, 0, 4u * nd); continue; }
            uint32_t *rec = slot + 8u * code;
            if (rec[1]) { /* the slot is taken: a new overflow record goes in behind the slot's own */
                uint32_t *o = ent + 8u * n_ent;
                o[0] = rec[0];
                rec[0] = ++n_ent;
                rec = o;
            }
            rec[1] = L;
            rec[2] = dw;
            const uint32_t tail = L < 16u ? L : 16u;
            memcpy((uint8_t *)(rec + 4) + (16u - tail), pat + (L - tail), tail);
            dw += nd;
        }
        suffix->sfx_slot_off = (uint32_t)base;
        suffix->sfx_ent_off = (uint32_t)(base + slot_bytes);
        suffix->sfx_pat_off = (uint32_t)(base + slot_bytes + ent_bytes);
        suffix->gram_bytes = (uint32_t)total;
    }
    free(suffix->gram_table);
    suffix->gram_kind = SMH_GRAM_PAIR2;
    suffix->gram_planes = 8;
    suffix->gram_jb = JB;
    suffix->gram_density = dens;
    suffix->gram_table = tab;
    suffix->gram_density = dens;
    return 0;
}

struct smh_wm *smh_wm_compile_impl(const unsigned char *pattern_flat, int m, int p_size, int alphabet,
                                   const int *SHIFT, const int *PREFIX_value, const int *PREFIX_index,
                                   const int *PREFIX_size)
{
    if (!pattern_flat || m < 3 || p_size < 1 || alphabet < 2 || alphabet > 256) {
        smh_set_error("smh_wm_compile: bad arguments (need m >= 3, p_size >= 1, 2 <= alphabet <= 256)");
        return NULL;
    }
    uint32_t ssz = smh_wu_shiftsize_for(alphabet);
    if (!ssz) {
        smh_set_error("The alphabet size is not supported by wu-manber");
        return NULL;
    }
    for (size_t i = 0; i < (size_t)m * p_size; ++i)
        if ((int)pattern_flat[i] >= alphabet) {
            smh_set_error("smh_wm_compile: symbol %u >= alphabet %d", pattern_flat[i], alphabet);
            return NULL;
        }
    struct smh_wm *wm = (struct smh_wm *)calloc(1, sizeof *wm);
    if (wm) wm->engine_forced = -1, wm->serial = smh_handle_serial();
    if (!wm) goto oom;
    wm->magic = SMH_MAGIC_WM;
    wm->alphabet = alphabet;
    wm->m = m;
    wm->patterns = p_size;
    wm->shiftsize = ssz;
    wm->bits_per_symbol = ceil_log2_u32((uint32_t)alphabet);

    /* ---- reference-layout tables ----
     * caller's dense tables (preproc_wu / preproc_wu2 output)  ->  CSR buckets, or
     * built here straight into CSR: same SHIFT values and the same bucket order
     * (pattern index ascending, wu/wu.c:140-143) without the dense
     * [shiftsize x p_size] arrays of main.c:436-439 */
    wm->l_shift = (int32_t *)malloc((size_t)ssz * sizeof(int32_t));
    wm->l_bucket_off = (uint32_t *)calloc((size_t)ssz + 1, sizeof(uint32_t));
    if (!wm->l_shift || !wm->l_bucket_off) goto oom;
    if (SHIFT) {
        uint64_t total = 0;
        for (uint32_t h = 0; h < ssz; ++h) {
            wm->l_shift[h] = SHIFT[h];
            if (PREFIX_size[h] < 0 || PREFIX_size[h] > p_size) {
                smh_set_error("smh_wm_compile_tables: PREFIX_size[%u] = %d out of range", h, PREFIX_size[h]);
                goto bad;
            }
            wm->l_bucket_off[h] = (uint32_t)total;
            total += (uint64_t)PREFIX_size[h];
        }
        wm->l_bucket_off[ssz] = (uint32_t)total;
        wm->l_bucket = (int32_t *)malloc((size_t)(total ? total : 1) * 2 * sizeof(int32_t));
        if (!wm->l_bucket) goto oom;
        for (uint32_t h = 0; h < ssz; ++h)
            for (int i = 0; i < PREFIX_size[h]; ++i) {
                size_t src = (size_t)h * p_size + (size_t)i;
                size_t dst = (size_t)wm->l_bucket_off[h] + (size_t)i;
                if (PREFIX_index[src] < 0 || PREFIX_index[src] >= p_size) {
                    smh_set_error("smh_wm_compile_tables: PREFIX_index out of range");
                    goto bad;
                }
                wm->l_bucket[2 * dst] = PREFIX_value[src];
                wm->l_bucket[2 * dst + 1] = PREFIX_index[src];
            }
    } else {
        const int nbits = 2, B = 3;
        for (uint32_t h = 0; h < ssz; ++h) wm->l_shift[h] = m - B + 1; /* main.c:444-449 */
        uint32_t *fillpos = (uint32_t *)calloc(ssz, sizeof(uint32_t));
        wm->l_bucket = (int32_t *)malloc((size_t)p_size * 2 * sizeof(int32_t));
        if (!fillpos || !wm->l_bucket) { free(fillpos); goto oom; }
        for (int j = 0; j < p_size; ++j) {
            const unsigned char *P = pattern_flat + (size_t)j * m;
            for (int end = m - 1; end >= B - 1; --end) {
                unsigned h = (((unsigned)P[end - 2] << nbits) + P[end - 1] << nbits) + P[end];
                if (m - 1 - end < wm->l_shift[h]) wm->l_shift[h] = m - 1 - end;
            }
            unsigned hs = (((unsigned)P[m - 3] << nbits) + P[m - 2] << nbits) + P[m - 1];
            wm->l_bucket_off[hs + 1]++;
        }
        for (uint32_t h = 0; h < ssz; ++h) wm->l_bucket_off[h + 1] += wm->l_bucket_off[h];
        for (int j = 0; j < p_size; ++j) {
            const unsigned char *P = pattern_flat + (size_t)j * m;
            unsigned hs = (((unsigned)P[m - 3] << nbits) + P[m - 2] << nbits) + P[m - 1];
            size_t dst = (size_t)wm->l_bucket_off[hs] + fillpos[hs]++;
            wm->l_bucket[2 * dst] = (int32_t)(((unsigned)P[0] << nbits) + P[1]);
            wm->l_bucket[2 * dst + 1] = j;
        }
        free(fillpos);
    }
    for (uint32_t h = 0; h < ssz; ++h)
        if (wm->l_shift[h] == 0) wm->shift_zero++;
    wm->pat_orig = (unsigned char *)malloc((size_t)p_size * m);
    if (!wm->pat_orig) goto oom;
    memcpy(wm->pat_orig, pattern_flat, (size_t)p_size * m);

    /* ---- distinct patterns, sorted ---- */
    wm->pat_sorted = (unsigned char *)malloc((size_t)p_size * m);
    if (!wm->pat_sorted) goto oom;
    memcpy(wm->pat_sorted, pattern_flat, (size_t)p_size * m);
    qsort_r(wm->pat_sorted, (size_t)p_size, (size_t)m, cmp_rows, &m);
    int d = 0;
    for (int j = 0; j < p_size; ++j)
        if (j == 0 || memcmp(wm->pat_sorted + (size_t)j * m, wm->pat_sorted + (size_t)(d - 1) * m, (size_t)m) != 0) {
            if (d != j) memmove(wm->pat_sorted + (size_t)d * m, wm->pat_sorted + (size_t)j * m, (size_t)m);
            ++d;
        }
    wm->distinct = d;

    /* ---- block filter (device SHIFT table, one bit per block code: 1 <=> SHIFT_dev == 0) ----
     * direct : block = last W symbols, W*bits <= 20, bit index = code
     * hashed : block = last min(m, 32/bits) symbols, two bits of one word chosen by a
     *          multiplicative hash (blocked Bloom filter, k = 2)
     * exact  : direct and W == m  ->  a set bit IS a match, no verify stage            */
    const int bits = wm->bits_per_symbol;
    int Wd = SMH_WM_FILTER_LOG2_MAX / bits;
    if (Wd > m) Wd = m;
    if (Wd < 1) Wd = 1;
    int Td = Wd * bits;
    if (Td < 5) Td = 5;
    /* density of the direct filter = distinct block codes / 2^Td */
    uint32_t *direct = (uint32_t *)calloc((size_t)1 << (Td - 5), sizeof(uint32_t));
    if (!direct) goto oom;
    uint64_t dset = 0;
    for (int j = 0; j < d; ++j) {
        uint32_t code = block_code(wm->pat_sorted + (size_t)j * m + (m - 1), Wd, bits);
        uint32_t w = code >> 5, b = 1u << (code & 31);
        if (!(direct[w] & b)) { direct[w] |= b; ++dset; }
    }
    /* ... over the codes a text CAN produce: 4 symbols of a 20-letter alphabet fill 160 000 of the 2^20 five-bit codes, so
     * 10 000 patterns pass 6 % of the columns, not the 0.95 % the bit count says (round 4: that set kept the direct filter,
     * 1.02 ms/GiB measured against 0.51 estimated) */
    double reach = 1.0;
    for (int i = 0; i < Wd; ++i) reach *= (double)alphabet;
    if (reach > (double)(1ull << Td)) reach = (double)(1ull << Td);
    double direct_density = (double)dset / reach;
    int exact = (Wd == m);
    int Wh = 32 / bits;
    if (Wh > m) Wh = m;
    int use_hashed = !exact && Wh > Wd && direct_density > 1.0 / 64.0;
    if (use_hashed) {
        const int Th = SMH_WM_FILTER_LOG2_MAX;
        const size_t nwords = (size_t)1 << (Th - 5);
        const int wbits = Wh * bits;
        const uint32_t kmask = wbits >= 32 ? 0xFFFFFFFFu : ((1u << wbits) - 1u);
        uint32_t *best = NULL;
        double best_density = 2.0;
        int best_k = 2;
        const int le4 = bits == 8 && Wh == 4;
        /* try 2, 3 and 4 bits per key (all inside one word, so the scan still costs one LDS lookup
         * per column) and keep the one that lets the fewest random keys through */
        double best_score = 1e30;
        for (int k = 2; k <= (le4 ? 5 : 4); ++k) {
            uint32_t *hashed = (uint32_t *)calloc(nwords, sizeof(uint32_t));
            if (!hashed) { free(direct); free(best); goto oom; }
            for (int j = 0; j < d; ++j) {
                uint32_t key = block_code(wm->pat_sorted + (size_t)j * m + (m - 1), Wh, bits) & kmask;
                if (le4) key = __builtin_bswap32(key); /* rolling code = big-endian; the scan reads the dword */
                if (le4) {
                    /* the byte-block form's own hash and bit layout: smh_wm_filter_key_v2 in wm_lane.h */
                    const uint32_t lo24 = key & 0xFFFFFFu;
                    const uint32_t h = (uint32_t)((uint64_t)lo24 * 0x9E3779u) + (key >> 24) * 0x85EBCBu;
                    const uint32_t g = (uint32_t)((uint64_t)lo24 * 0xC2B2AFu);
                    const uint32_t blk = (h >> 3) & 0x3FFFu;
                    hashed[2u * blk] |= (1u << ((g >> 8) & 31u)) | (1u << ((g >> 16) & 31u));
                    if (k >= 3) hashed[2u * blk + 1u] |= 1u << ((g >> 24) & 31u);
                    if (k >= 4) hashed[2u * blk + 1u] |= 1u << (g & 31u);
                    if (k >= 5) hashed[2u * blk] |= 1u << ((h >> 24) & 31u);
                    continue;
                }
                uint32_t h = smh_wm_block_hash(key);
                /* 64-bit blocks, bit positions: smh_wm_filter_key in wm_lane.h */
                uint32_t blk = h >> (32 - (Th - 6));
                hashed[2u * blk] |= (1u << (h & 31u)) | (1u << ((h >> 5) & 31u));
                if (k >= 3) hashed[2u * blk + 1u] |= 1u << ((h >> 10) & 31u);
                if (k >= 4) hashed[2u * blk + 1u] |= 1u << ((h >> 13) & 31u);
            }
            /* pass probability of a random key ~ mean over blocks of f_lo^2 * f_hi^(k-2) */
            double acc = 0;
            for (size_t b = 0; b < nwords / 2; ++b) {
                const double fl = (double)__builtin_popcount(hashed[2 * b]) / 32.0;
                const double fh = (double)__builtin_popcount(hashed[2 * b + 1]) / 32.0;
                double pk = fl * fl;
                for (int i = 2; i < k && i < 4; ++i) pk *= fh;
                if (k >= 5) pk *= fl; /* the byte-block form's fifth bit is in the low dword again */
                acc += pk;
            }
            const double density = acc / (double)(nwords / 2);
            /* the byte-block form pays two VALU per bit and column (0.03 ms/GiB) and the staged verify per survivor:
             * the cheapest total wins, not the sparsest filter; the other forms keep the sparsest */
            const double score = le4 && m <= 33 ? 0.27 + 0.032 * k + gram_verify_ms(m, density) : density;
            if (score < best_score) {
                free(best);
                best = hashed;
                best_density = density;
                best_score = score;
                best_k = k;
            } else {
                free(hashed);
            }
        }
        if (best_density < direct_density) {
            free(direct);
            wm->filter = best;
            wm->filter_hashed = 1;
            wm->filter_k = best_k;
            wm->filter_le4 = le4;
            wm->filter_log2 = Th;
            wm->block_symbols = Wh;
            wm->filter_density = best_density;
        } else {
            free(best);
            use_hashed = 0;
        }
    }
    if (!use_hashed) {
        wm->filter = direct;
        wm->filter_hashed = 0;
        wm->filter_log2 = Td;
        wm->block_symbols = Wd;
        wm->filter_density = direct_density;
        wm->filter_exact = exact;
    }

    /* ---- pair filter: one lookup for two end columns (see smh_internal.h) ---- */
    if (wm->filter_exact && alphabet == 4 && m <= 8 && !wm->filter_hashed) {
        wm->pair_table = (uint32_t *)calloc(16384u, sizeof(uint32_t));
        if (!wm->pair_table) goto oom;
        const uint32_t mmask = (1u << (2 * m)) - 1u;
        for (uint32_t i = 0; i < (1u << 18); ++i) {
            /* the exact direct filter is indexed by the m-symbol code, oldest symbol highest */
            const uint32_t c1 = (i >> 2) & mmask, c2 = i & mmask, pair = i & 15u;
            if ((wm->filter[c1 >> 5] >> (c1 & 31u)) & 1u) wm->pair_table[i >> 4] |= 1u << (2u * pair);
            if ((wm->filter[c2 >> 5] >> (c2 & 31u)) & 1u) wm->pair_table[i >> 4] |= 1u << (2u * pair + 1u);
        }
    }

    /* ---- verify table (device HASH/PREFIX stage): hash(window) -> one 32-bit slot, 12 tag bits above
     *      (pattern + 1) in 20 bits, 0 = empty.  Four bytes a slot keep the table of 100 000 patterns at
     *      1 MiB, so the random probes of the verify stage mostly hit L2 beside the streaming text ---- */
    {   /* built for exact filters too: a handle may serve as one length class of a mixed-length set */
        if (d >= (1 << 20) - 1) {
            smh_set_error("smh_wm_compile: more than 2^20 - 2 distinct patterns");
            goto bad;
        }
        /* four slots per pattern or more (2 MiB at 100 000 patterns): a bucket is then full -- and a probe needs a second,
         * un-pipelined round trip -- for 1 % of the windows; at two per pattern it was 7-14 % of them, which showed as
         * half of the verify stage's time once the first bucket's load was software-pipelined (gpurun_out/r02_aa) */
        int lg = ceil_log2_u32((uint32_t)d * 4u);
        /* ... while the table stays within 2 MiB.  (Rounds 2-3 capped it at 1 MiB: beside the streaming text a 2 MiB table no
         * longer lives in a 4 MiB L2 -- 100 000 patterns 6 % faster with it then, at 1.32x the algorithmic HBM traffic instead of
         * 1.08x.  Round 4: with the windows from L2 and the first bucket's request pipelined across chunks, a full bucket's second,
         * un-pipelined trip is what is left exposed -- 2.6 slots per pattern: 7 % of the windows, four: 1 % -- and the same
         * 2 MiB table is 8-11 % faster, 100 000 byte patterns over 4 GiB 1.53 / 1.26 / 1.20 / 1.21 -> 1.42 / 1.15 / 1.08 / 1.07 ms
         * at m = 5 / 8 / 12 / 20; 4 MiB: no better.  SMH_WM_TUNE="vt=1m" restores the old cap.) */
        const size_t vt_cap = smh_tune_has(SMH_TUNE_WM, "vt=1m") ? (size_t)1 << 20 : (size_t)2 << 20;
        if (((size_t)4 << lg) > vt_cap) lg = ceil_log2_u32((uint32_t)d * 2u);
        if (lg < 4) lg = 4;
        wm->verify_log2 = lg;
        size_t slots = (size_t)1 << lg;
        wm->verify = (uint32_t *)calloc(slots, sizeof(uint32_t));
        if (!wm->verify) goto oom;
        /* 16-byte buckets of four slots, filled in order: one 16-byte load shows a probe all four tags, and a
         * bucket whose last slot is empty ends an unsuccessful search (linear probing over single slots needed
         * up to ten dependent loads for the unluckiest of the 128 columns of a drain) */
        const size_t nb = slots / 4;
        for (int j = 0; j < d; ++j) {
            uint32_t tag = smh_wm_tag(wm->pat_sorted + (size_t)j * m, m);
            size_t b = (size_t)((tag * SMH_HASH_MUL) >> (32 - (lg - 2)));
            for (;;) {
                uint32_t *q = wm->verify + 4 * b;
                int k = 0;
                while (k < 4 && q[k]) ++k;
                if (k < 4) { q[k] = ((tag & 0xFFFu) << 20) | ((uint32_t)j + 1u); break; }
                b = (b + 1) & (nb - 1);
            }
        }
    }

    /* ---- the verify entries once more as a cuckoo hash (smh_internal.h verify_ck): sets whose table above exceeds 512 KiB ---- */
    if (((size_t)4 << wm->verify_log2) > ((size_t)512 << 10)) {
        const uint32_t NB = (uint32_t)((double)d / (4.0 * 0.82)) + 4u;
        uint32_t *ck = (uint32_t *)calloc(4u * (size_t)NB, sizeof(uint32_t));
        if (!ck) goto oom;
        int ok = 0;
        uint64_t rng = 0x9E3779B97F4A7C15ull;
        uint32_t seed = 0;
        for (uint32_t attempt = 0; attempt < 16u && !ok; ++attempt) {
            seed = attempt * 0x7F4A7C15u;
            memset(ck, 0, sizeof(uint32_t) * 4u * (size_t)NB);
            ok = 1;
            for (int j = 0; j < d && ok; ++j) {
                uint32_t cur = (uint32_t)j + 1u; /* entries are placed as pattern numbers and encoded below */
                int done = 0;
                for (uint32_t kicks = 0; kicks < 4000u && !done; ++kicks) {
                    uint32_t b1, b2;
                    smh_hash_slots(smh_wm_tag(wm->pat_sorted + (size_t)(cur - 1u) * m, m), seed, NB, &b1, &b2);
                    const uint32_t cand[4] = {2u * b1, 2u * b1 + 1u, 2u * b2, 2u * b2 + 1u};
                    for (int c = 0; c < 4 && !done; ++c)
                        if (!ck[cand[c]]) { ck[cand[c]] = cur; done = 1; }
                    if (done) break;
                    rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                    const uint32_t victim = cand[(rng >> 33) & 3u], out = ck[victim];
                    ck[victim] = cur;
                    cur = out;
                }
                ok = done;
            }
        }
        if (ok) {
            for (uint32_t i = 0; i < 4u * NB; ++i)
                if (ck[i]) ck[i] = ((smh_wm_tag(wm->pat_sorted + (size_t)(ck[i] - 1u) * m, m) & 0xFFFu) << 20) | ck[i];
            wm->verify_ck = ck;
            wm->ck_buckets = NB;
            wm->ck_seed = seed;
        } else {
            free(ck); /* the bucket table above serves */
        }
    }

    /* ---- gram filter: taken when this path would otherwise scan with a NON-exact filter and the gram filter
     *      lets fewer columns through (it also costs less per column: one lookup per two columns on the
     *      4-letter alphabet, no hash arithmetic beyond one 24-bit multiply on byte symbols) ---- */
    if (!wm->filter_exact && !wm->pair_table) {
        /* this path's block filter, same units: a non-exact direct filter scans at 0.40 ms/GiB, the hashed
         * byte-block filter at 0.55, and their survivors cost the same verify stage */
        /* the byte-block form (12 VALU per column, staged verify): 0.65 ms/GiB at 100 000 patterns of 5 bytes, 1.8 % passing */
        const double other_ms = wm->filter_hashed && wm->filter_le4 && m <= 33 ? 0.27 + 0.032 * wm->filter_k + gram_verify_ms(m, wm->filter_density)
                              : wm->filter_hashed ? 0.55 + SMH_HASHED_VERIFY_MS(m) * wm->filter_density
                                                  : 0.40 + SMH_DIRECT_VERIFY_MS(m) * wm->filter_density;
        wm->scan_ms_est = other_ms;
        if (build_gram_filter(wm, other_ms) != 0) goto oom;
    }

    /* Scan-engine choice, the mirror image of the one in ac_host.c: a small-alphabet set of LONG patterns
     * leaves this path with a non-exact direct filter -- one LDS lookup per column plus the verify stage,
     * 2.4-2.6 TB/s -- while the automaton kernels, when the set's automaton fits LDS with next to no
     * candidates, run it at 3.5-3.6 TB/s.  Same count either way. */
    /* (alphabets 9..32 only for the sake of the text-independent parts: not when the set cannot fit SMH_FLAT_MAX_PARTS of them
     * even if its patterns shared three symbols in four) */
    const uint64_t flat_rows_cap = (uint64_t)SMH_FLAT_MAX_PARTS * (SMH_AC_LDS_BUDGET / ((uint64_t)alphabet * 2u));
    if (alphabet <= 32 && (alphabet <= 8 || (uint64_t)d * (uint64_t)m <= 4u * flat_rows_cap) && !wm->filter_exact && !wm->pair_table &&
        smh_alt_engine_depth == 0) {
        ++smh_alt_engine_depth;
        struct smh_ac *ac = smh_ac_compile_patterns(wm->pat_sorted, m, d, alphabet);
        --smh_alt_engine_depth;
        /* the automaton plan's cost is in units of the exact stride-1 scan, 0.289 ms/GiB */
        if (ac && ac->scan_cost <= SMH_WM_ALT_ENGINE_COST && ac->scan_cost * 0.289 < wm->scan_ms_est)
            wm->alt_ac = ac;
        /* round 4: an automaton that is merely SLOWER on random text stays at hand -- a filter's speed is a property of the
         * text (survivors are verified one by one), the automaton's next to none, and the runtime follows what the launches
         * report (smh_runtime.hip "adaptive engine") */
        /* ... and so does one that is out of the race itself but brought the text-independent engine along (the set as a few
         * exact stride-1 automata, ac_host.c): that one is the floor under both */
        if (ac && (wm->alt_ac == ac || (ac->fixed_length_ok && (ac->scan_cost <= SMH_WM_FLEX_ENGINE_COST || ac->flat_ac))))
            wm->flex_ac = ac;
        else
            smh_ac_free(ac);
    }
    /* (round 4) 4-letter sets of 9 or 10 symbols have the EXACT direct filter -- one LDS lookup per column, no verify stage,
     * 0.28-0.32 ms/GiB whatever the set (tools/est_check.py) -- and no pair table (its index would be 2^20 / 2^22 entries).  While
     * the set is small enough for an exact automaton image the automaton kernels run it at 0.16-0.26: same mirror-image choice
     * as above, static (both engines are exact: nothing about the text could change it) */
    if (alphabet == 4 && wm->filter_exact && !wm->pair_table && !wm->gram_table && m >= 9 && smh_alt_engine_depth == 0) {
        wm->scan_ms_est = 0.30;
        ++smh_alt_engine_depth;
        struct smh_ac *ac = smh_ac_compile_patterns(wm->pat_sorted, m, d, alphabet);
        --smh_alt_engine_depth;
        if (ac && ac->fixed_length_ok && ac->scan_exact && !ac->scan_full_rows && smh_ac_plan_ms(ac) + 0.02 < wm->scan_ms_est) {
            wm->alt_ac = ac; /* not flex_ac: nothing to adapt between two exact engines */
        } else {
            smh_ac_free(ac);
        }
    }
    /* Round 5: the key engine (key_hash.h) beside every path whose rate depends on the text (a filter with a verify stage) */
    if (!wm->filter_exact && !wm->pair_table && smh_alt_engine_depth == 0 && m * smh_keys_symbol_bits(alphabet) <= SMH_KEY_MAX_BITS &&
        !(wm->flex_ac && wm->flex_ac->flat_parts == 1))
        wm->keys = smh_keys_build(wm->pat_sorted, m, d, alphabet, SMH_KEYS_LDS_BUDGET, NULL);
    /* ... and where the key engine does not take the set (more keys than LDS holds, m * bits > 64), byte-like alphabets get the
     * window-hash engine (hash_engine.h): a filter whose pass rate does not depend on the text either */
    /* (also for the handle an automaton handle keeps as its filter engine -- depth 1 --: that handle runs it as ITS fifth engine) */
    if (!wm->keys && !wm->filter_exact && !wm->pair_table && smh_alt_engine_depth <= 1 && alphabet > 4 && m >= SMH_HASH_MIN_M && m <= SMH_HASH_MAX_M)
        wm->hashes = smh_hash_build(wm->pat_sorted, m, d, NULL);
    return wm;

oom:
    smh_set_error("smh_wm_compile: out of memory");
bad:
    smh_wm_host_free(wm);
    return NULL;
}

smh_wm *smh_wm_compile(const unsigned char *pattern_flat, int m, int p_size, int alphabet)
{
    return smh_wm_compile_impl(pattern_flat, m, p_size, alphabet, NULL, NULL, NULL, NULL);
}

smh_wm *smh_wm_compile_tables(const unsigned char *pattern_flat, int m, int p_size, int alphabet,
                              const int *SHIFT, const int *PREFIX_value, const int *PREFIX_index,
                              const int *PREFIX_size)
{
    if (!SHIFT || !PREFIX_value || !PREFIX_index || !PREFIX_size) {
        smh_set_error("smh_wm_compile_tables: NULL table");
        return NULL;
    }
    return smh_wm_compile_impl(pattern_flat, m, p_size, alphabet, SHIFT, PREFIX_value, PREFIX_index,
                               PREFIX_size);
}

int smh_wm_get_info(const smh_wm *wm, smh_wm_info *out)
{
    if (!wm || wm->magic != SMH_MAGIC_WM || !out) {
        smh_set_error("smh_wm_get_info: bad handle");
        return SMH_EINVAL;
    }
    memset(out, 0, sizeof *out);
    out->alphabet = (uint32_t)wm->alphabet;
    out->m = (uint32_t)wm->m;
    out->patterns = (uint32_t)wm->patterns;
    out->distinct = (uint32_t)wm->distinct;
    out->shiftsize = wm->shiftsize;
    out->shift_zero = wm->shift_zero;
    out->block_symbols = (uint32_t)wm->block_symbols;
    out->filter_log2 = (uint32_t)wm->filter_log2;
    out->filter_exact = (uint32_t)wm->filter_exact;
    out->filter_hashed = (uint32_t)wm->filter_hashed;
    out->verify_slots = wm->filter_exact ? 0u : (1u << wm->verify_log2);
    out->lds_bytes = (uint32_t)(((size_t)1 << wm->filter_log2) / 8);
    out->scan_engine = wm->engine_forced >= 0 ? (uint32_t)wm->engine_forced : (wm->alt_ac ? SMH_ALGO_AC : SMH_ALGO_WM);
    out->adaptive = (wm->flex_ac || wm->keys || wm->hashes) && wm->engine_forced < 0 ? 1u : 0u;
    out->key_slots = wm->keys ? 2u * wm->keys->P.slots : 0u;
    out->hash_slots = wm->hashes ? 4u * wm->hashes->P.slots : 0u; /* two tables of two-slot buckets */
    out->verify_ck_slots = wm->verify_ck ? 4u * wm->ck_buckets : 0u;
    out->gram_planes = wm->gram_kind != SMH_GRAM_NONE ? (uint32_t)wm->gram_planes : 0u;
    out->gram_kind = (uint32_t)wm->gram_kind;
    { /* which verify stage a launch over text like the compile's takes (wm_kernels.inc launch_gram): 1 in registers, 2 windows from L2, 0 staged / other */
        const double pc = wm->gram_density * 4096.0;
        const int pairlike = wm->gram_kind == SMH_GRAM_PAIR || wm->gram_kind == SMH_GRAM_OCT2, dna = pairlike || wm->gram_kind == SMH_GRAM_OCT;
        out->verify_in_registers = 0;
        if (dna && wm->m <= 33 && pc >= (pairlike ? SMH_L2_MIN_PER_CHUNK_REGV : SMH_L2_MIN_PER_CHUNK) && pc <= SMH_L2_DNA_MAX_PER_CHUNK) out->verify_in_registers = 2;
        else if (pairlike && wm->m <= 33 && SMH_REGV_WANTED(pc)) out->verify_in_registers = 1;
    }
    if (wm->gram_kind != SMH_GRAM_NONE) out->lds_bytes = wm->gram_kind == SMH_GRAM_OCT ? 65536u : (wm->gram_kind == SMH_GRAM_BYTE_BIG || wm->gram_kind == SMH_GRAM_FLAT_BIG || wm->gram_kind == SMH_GRAM_FLAT4_BIG ? SMH_GRAM_BIG_BYTES : SMH_GRAM_BYTES);
    return SMH_OK;
}

int smh_wm_set_scan_engine(smh_wm *wm, int engine)
{
    if (engine == SMH_ENGINE_AC_FLAT && wm && wm->magic == SMH_MAGIC_WM && !(wm->flex_ac && wm->flex_ac->flat_ac)) {
        smh_set_error("smh_wm_set_scan_engine: this set keeps no plain stride-1 automaton");
        return SMH_EUNSUP;
    }
    if (!wm || wm->magic != SMH_MAGIC_WM || (engine != -1 && engine != SMH_ALGO_WM && engine != SMH_ALGO_AC && engine != SMH_ENGINE_AC_FLAT && engine != SMH_ENGINE_KEYS && engine != SMH_ENGINE_HASH)) {
        smh_set_error("smh_wm_set_scan_engine: bad arguments");
        return SMH_EINVAL;
    }
    if (engine == SMH_ENGINE_HASH && !wm->hashes) {
        smh_set_error("smh_wm_set_scan_engine: this handle keeps no window-hash engine (it keeps a key table, its path is exact, or m is outside 4..32)");
        return SMH_EUNSUP;
    }
    if (engine == SMH_ENGINE_KEYS && !wm->keys) {
        smh_set_error("smh_wm_set_scan_engine: this handle keeps no key table (m * bits per symbol > 64, more keys than LDS holds, or this path is exact)");
        return SMH_EUNSUP;
    }
    if (engine == SMH_ALGO_AC && !wm->alt_ac && !wm->flex_ac) {
        smh_set_error("smh_wm_set_scan_engine: this set has no automaton engine (its automaton would be slower)");
        return SMH_EUNSUP;
    }
    wm->alt_off = engine == SMH_ALGO_WM;
    wm->engine_forced = engine;
    ++wm->generation;
    return SMH_OK;
}

void smh_wm_free(smh_wm *wm)
{
    if (!wm) return;
    if (wm->dev) smh_wm_dev_free(wm->dev);
    wm->dev = NULL;
    smh_adapt_dev_free(wm->adapt);
    wm->adapt = NULL;
    smh_wm_host_free(wm);
}
