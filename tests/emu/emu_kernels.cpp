/*
 * tests/emu/emu_kernels.cpp -- TEST HARNESS ONLY, never linked into the product.
 *
 * Compiles the kernels' lane code (csrc/ac_lane.h, csrc/wm_lane.h) for the CPU
 * with -DSMH_HOST_EMU and runs it thread by thread over the same grid the
 * launchers would use.  This is how tiling, halo, tail and early-exit logic is
 * exercised in the GPU-less authoring container; the -m gpu tests then run
 * the real kernels through the C ABI.  The text is copied into a buffer whose
 * last valid 16-byte piece ends at a PROT_NONE guard page (and, in a second
 * pass, starts right after one), so an out-of-bounds vector load in a fast
 * path crashes the test instead of passing silently.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>
#include <vector>
#include "smh_internal.h"
#include "ac_lane.h"
#include "wm_lane.h"

/* the kernels' cold context (ac_lane.h smh_ac_cold_ctx) with host pointers: full DFA, stride-1 table, depth_first, and the hash
 * verify of the automaton kernels (ac_host.c hv_wm; patterns zero-padded to whole dwords) */
struct emu_hv {
    std::vector<uint8_t> padded;
    smh_ac_cold_ctx C;
    void fill(const smh_ac *ac, smh_ac_verify_ctx &V, const uint32_t *depth_first)
    {
        C = smh_ac_cold_ctx{};
        C.full = ac->table; C.full_entry_bytes = ac->entry_bytes; C.depth_first = depth_first;
        C.trunc1 = ac->trunc1_table; C.trunc1_entry_bytes = ac->trunc1_entry_bytes;
        V.cold = &C;
        const smh_wm *w = ac->hv_wm;
        if (!w || !w->verify) return;
        const size_t row = (size_t)((w->m + 3) / 4) * 4;
        padded.assign((size_t)w->distinct * row + 16, 0);
        for (int j = 0; j < w->distinct; ++j) memcpy(padded.data() + (size_t)j * row, w->pat_sorted + (size_t)j * w->m, (size_t)w->m);
        C.hv_verify = w->verify; C.hv_pats = padded.data(); C.hv_log2 = w->verify_log2;
    }
};


#define EMU_BLOCK_THREADS 1024
static int emu_tune(const char *key, int dflt)
{
    const char *t = getenv("SMH_AC_TUNE");
    if (!t) return dflt;
    const char *p = strstr(t, key);
    if (!p) return dflt;
    return atoi(p + strlen(key) + 1);
}
#define EMU_AC_NCH 1 /* == SMH_AC_NCH in ac_kernels.hip */

struct guarded {
    uint8_t *map;
    size_t map_len;
    uint8_t *text;
};

/* mode 0: buffer ends at a guard page; mode 1: buffer starts right after one */
static guarded guard_copy(const uint8_t *src, uint64_t n, int mode)
{
    const size_t page = (size_t)sysconf(_SC_PAGESIZE);
    const size_t padded = ((n + 15) / 16) * 16; /* the runtime pads device text to 16 bytes */
    const size_t body = ((padded + page - 1) / page) * page;
    guarded g;
    g.map_len = body + 2 * page;
    g.map = (uint8_t *)mmap(NULL, g.map_len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (g.map == MAP_FAILED) abort();
    mprotect(g.map, page, PROT_NONE);
    mprotect(g.map + page + body, page, PROT_NONE);
    g.text = mode == 0 ? g.map + page + body - padded : g.map + page;
    memcpy(g.text, src, n);
    return g;
}

static void guard_free(guarded &g) { munmap(g.map, g.map_len); }

/* ------------------------------------------------------------------ AC */
template <typename E, int SIGMA, int STRIDE, int HC, bool EXACT, bool POS = false>
static uint64_t ac_grid(const smh_ac *ac, const smh_ac_verify_ctx &V, uint64_t blocks)
{
    if constexpr (POS && HC > 2) return ~0ull; /* as the launcher: positions mode covers a 32-byte halo */
    else {
    smh_ac_df df;
    for (int i = 0; i < SMH_AC_DF_LEN; ++i) df.v[i] = i <= ac->max_depth + 1 ? ac->depth_first[i] : ac->rows;
    if (STRIDE == 3) /* as smh_runtime.hip: compact rows of the hybrid image count as "deep enough" */
        for (int i = 0; i < SMH_AC_DF_LEN; ++i)
            if (df.v[i] > ac->scan_full_rows) df.v[i] = ac->scan_full_rows;
    const uint64_t nthreads = blocks * EMU_BLOCK_THREADS;
    uint64_t total = 0;
    for (uint64_t t = 0; t < nthreads; ++t) {
        if constexpr (STRIDE == 3) {
            /* as launch_chains<3> / launch_hybrid_chains (ac_kernels.inc): the multi-chain instantiations run the UNCLAMPED
             * format (the out-of-range full-row read modelled as 0, what the device probe guarantees), exact plans with two
             * chains and the register prefetch, depth-cut plans and positions mode with three; SMH_AC_TUNE="clamp=1" and
             * the one-chain instantiations (halo beyond 16 bytes) the clamped one */
            if constexpr (HC == 1) {
                const bool two = EXACT && !POS && emu_tune("nch", 2) == 2;
                if (emu_tune("clamp", 0) == 0) {
                    const smh_fmt_s2h_oob fmt = smh_fmt_s2h_oob::make(ac->scan_full_rows, ac->scan_bytes);
                    total += two ? smh_ac_thread<smh_fmt_s2h_oob, 1, 2, EXACT, 1, 16, POS>(fmt, t, smh_sched_static(t >> 6, nthreads >> 6), ac->scan_table, V, df, nullptr)
                                 : smh_ac_thread<smh_fmt_s2h_oob, 1, 3, EXACT, 0, 16, POS>(fmt, t, smh_sched_static(t >> 6, nthreads >> 6), ac->scan_table, V, df, nullptr);
                } else {
                    const smh_fmt_s2h fmt = smh_fmt_s2h::make(ac->scan_full_rows, ac->scan_bytes);
                    total += two ? smh_ac_thread<smh_fmt_s2h, 1, 2, EXACT, 1, 16, POS>(fmt, t, smh_sched_static(t >> 6, nthreads >> 6), ac->scan_table, V, df, nullptr)
                                 : smh_ac_thread<smh_fmt_s2h, 1, 3, EXACT, 0, 16, POS>(fmt, t, smh_sched_static(t >> 6, nthreads >> 6), ac->scan_table, V, df, nullptr);
                }
            } else if constexpr (EXACT || HC <= 2) {
                const smh_fmt_s2h fmt = smh_fmt_s2h::make(ac->scan_full_rows, ac->scan_bytes);
                total += smh_ac_thread<smh_fmt_s2h, HC, EMU_AC_NCH, EXACT, SMH_PREFETCH, 16, POS>(fmt, t, smh_sched_static(t >> 6, nthreads >> 6), ac->scan_table, V, df, nullptr);
            }
        } else if constexpr (STRIDE == 2) {
            total += smh_ac_thread<smh_fmt_s2, HC, EMU_AC_NCH, EXACT, SMH_PREFETCH, 16, POS>(smh_fmt_s2{}, t, smh_sched_static(t >> 6, nthreads >> 6), ac->scan_table, V, df, nullptr);
        } else {
            const smh_fmt_s1<E, SIGMA> fmt{V.sigma};
            total += smh_ac_thread<smh_fmt_s1<E, SIGMA>, HC, EMU_AC_NCH, EXACT, SMH_PREFETCH, 16, POS>(fmt, t, smh_sched_static(t >> 6, nthreads >> 6), ac->scan_table, V, df, nullptr);
        }
    }
    return total;
    }
}

template <typename E, int SIGMA, int STRIDE, int HC, bool POS = false>
static uint64_t ac_exact(const smh_ac *ac, const smh_ac_verify_ctx &V, uint64_t blocks)
{
    return ac->scan_exact ? ac_grid<E, SIGMA, STRIDE, HC, true, POS>(ac, V, blocks)
                          : ac_grid<E, SIGMA, STRIDE, HC, false, POS>(ac, V, blocks);
}

template <typename E, int SIGMA, int STRIDE, bool POS = false>
static uint64_t ac_halo(const smh_ac *ac, const smh_ac_verify_ctx &V, uint64_t blocks)
{
    const int halo = ac->scan_depth - 1;
    if (halo <= 16) return ac_exact<E, SIGMA, STRIDE, 1, POS>(ac, V, blocks);
    if (halo <= 32) return ac_exact<E, SIGMA, STRIDE, 2, POS>(ac, V, blocks);
    return ac_exact<E, SIGMA, STRIDE, 4, POS>(ac, V, blocks);
}

/* the tuned kernels in positions mode, whatever scan plan the handle holds; ~0 = the plan's halo is
 * beyond the 32 bytes the mode covers (smh_ac_positions then runs the per-segment kernel) */
extern "C" uint64_t emu_ac_positions_tuned(const smh_ac *ac, const uint8_t *text, uint64_t n, uint64_t *out,
                                           uint64_t capacity, uint32_t blocks)
{
    if (n < (uint64_t)ac->m) return 0;
    if (!blocks) blocks = 2;
    size_t dflen = (size_t)ac->m + 2 < 72 ? 72 : (size_t)ac->m + 2;
    std::vector<uint32_t> df(dflen);
    for (size_t i = 0; i < dflen; ++i) df[i] = (int)i <= ac->max_depth + 1 ? ac->depth_first[i] : ac->rows;
    /* the fast path reads 16-byte pieces: give the text the padding the runtime gives device text */
    std::vector<uint8_t> padded(((n + 15) / 16) * 16 + 64, 0);
    memcpy(padded.data(), text, n);
    smh_ac_verify_ctx V = {};
    V.text = padded.data(); V.n = n; V.m = ac->m; V.K = ac->scan_depth; V.sigma = ac->alphabet;
    emu_hv hv;
    hv.fill(ac, V, df.data());
    uint64_t cursor = 0;
    V.pos = smh_pos_out{out, capacity, &cursor};
    uint64_t r;
    if (ac->scan_stride == 2 && ac->scan_full_rows)
        r = ac_halo<uint16_t, 4, 3, true>(ac, V, blocks);
    else if (ac->scan_stride == 2)
        r = ac_halo<uint16_t, 4, 2, true>(ac, V, blocks);
    else if (ac->scan_entry_bytes == 2)
        r = ac->alphabet == 4 ? ac_halo<uint16_t, 4, 1, true>(ac, V, blocks) : ac_halo<uint16_t, 0, 1, true>(ac, V, blocks);
    else
        r = ac->alphabet == 4 ? ac_halo<uint32_t, 4, 1, true>(ac, V, blocks) : ac_halo<uint32_t, 0, 1, true>(ac, V, blocks);
    return r == ~0ull ? ~0ull : cursor;
}

/* blocks: grid size (0 = 4); the scan plan (stride, depth K) is whatever the handle holds */
extern "C" uint64_t emu_ac_scan(const smh_ac *ac, const uint8_t *text_in, uint64_t n, int variant, uint32_t blocks)
{
    if (n < (uint64_t)ac->m) return 0;
    if (!blocks) blocks = 4;
    size_t dflen = (size_t)ac->m + 2 < 72 ? 72 : (size_t)ac->m + 2;
    uint32_t *df = (uint32_t *)malloc(dflen * 4);
    for (size_t i = 0; i < dflen; ++i) df[i] = (int)i <= ac->max_depth + 1 ? ac->depth_first[i] : ac->rows;
    uint64_t result[2];
    for (int mode = 0; mode < 2; ++mode) {
        guarded g = guard_copy(text_in, n, mode);
        const uint8_t *text = g.text;
        uint64_t total = 0;
        if (variant == SMH_VARIANT_TABLE) {
            const uint64_t nthreads = (uint64_t)blocks * 256;
            for (uint64_t t = 0; t < nthreads; ++t)
                total += smh_ac_table_thread(t, nthreads, text, n, ac->m, ac->g_transition, ac->g_supply,
                                             ac->g_final, ac->alphabet);
        } else if (ac->scan_dense) { /* the dense plan: as smh_ac_scan, the pair lane code over the automaton's accepting bits */
            const uint64_t nthreads = (uint64_t)blocks * EMU_BLOCK_THREADS;
            for (uint64_t t = 0; t < nthreads; ++t)
                total += smh_wm_pair_thread<SMH_PREFETCH != 0>(t, smh_sched_static(t >> 6, nthreads >> 6), text, n, ac->m, ac->dense_pair, ac->dense_filter);
        } else {
            smh_ac_verify_ctx V = {};
            V.text = text; V.n = n; V.m = ac->m; V.K = ac->scan_depth; V.sigma = ac->alphabet;
            emu_hv hv;
            hv.fill(ac, V, df);
            if (ac->scan_stride == 2 && ac->scan_full_rows)
                total = ac_halo<uint16_t, 4, 3>(ac, V, blocks);
            else if (ac->scan_stride == 2)
                total = ac_halo<uint16_t, 4, 2>(ac, V, blocks);
            else if (ac->scan_entry_bytes == 2)
                total = ac->alphabet == 4 ? ac_halo<uint16_t, 4, 1>(ac, V, blocks) : ac_halo<uint16_t, 0, 1>(ac, V, blocks);
            else
                total = ac->alphabet == 4 ? ac_halo<uint32_t, 4, 1>(ac, V, blocks) : ac_halo<uint32_t, 0, 1>(ac, V, blocks);
        }
        guard_free(g);
        result[mode] = total;
    }
    free(df);
    return result[0] == result[1] ? result[0] : ~0ull;
}

/* ------------------------------------------------------------------ WM */
/* mixed-length one-pass emulation: when set, wm_grid verifies against these classes (host pointers) */
static const smh_wm_class *g_emu_classes = nullptr;
static int g_emu_n_classes = 0;

template <bool HASHED, bool EXACT, int HC, int FK = 0, bool POS = false, bool STG = false>
static uint64_t wm_grid(const smh_wm *wm, const uint8_t *text, uint64_t n, uint64_t blocks, const smh_pos_out *po = nullptr)
{
    smh_wm_params P = {};
    P.m = wm->m;
    P.bits = wm->bits_per_symbol;
    const int wbits = wm->block_symbols * wm->bits_per_symbol;
    P.code_mask = wbits >= 32 ? 0xFFFFFFFFu : ((1u << wbits) - 1u);
    P.filter_log2 = wm->filter_log2;
    P.filter_k = wm->filter_k;
    P.filter_le4 = wm->filter_le4;
    P.n_classes = g_emu_n_classes;
    P.classes = g_emu_classes;
    P.verify_log2 = wm->verify_log2;
    P.verify = wm->verify;
    P.verify_ck = wm->verify_ck; P.ck_buckets = wm->ck_buckets; P.ck_seed = wm->ck_seed;
    /* distinct patterns zero-padded to whole dwords, as smh_runtime.hip uploads them */
    const size_t row = (size_t)((wm->m + 3) / 4) * 4;
    std::vector<uint8_t> padded((size_t)wm->distinct * row + 16, 0);
    for (int j = 0; j < wm->distinct; ++j) memcpy(padded.data() + (size_t)j * row, wm->pat_sorted + (size_t)j * wm->m, (size_t)wm->m);
    P.pat_sorted = padded.data();
    const uint64_t nthreads = blocks * EMU_BLOCK_THREADS;
    uint64_t total = 0;
    for (uint64_t t = 0; t < nthreads; ++t)
        total += smh_wm_thread<HASHED, EXACT, HC, FK, POS, STG>(t, smh_sched_static(t >> 6, nthreads >> 6), text, n, wm->filter, P, wm->block_symbols, nullptr, po);
    return total;
}

/* gram filter (q-gram shift-or): same lane code, true inherited states instead of the DPP correction */
static uint64_t wm_gram_grid(const smh_wm *wm, const uint8_t *text, uint64_t n, uint64_t blocks, const smh_pos_out *po)
{
    smh_wm_params P = {};
    P.m = wm->m;
    P.bits = wm->bits_per_symbol;
    P.verify_log2 = wm->verify_log2;
    P.verify = wm->verify;
    P.verify_ck = wm->verify_ck; P.ck_buckets = wm->ck_buckets; P.ck_seed = wm->ck_seed;
    const size_t row = (size_t)((wm->m + 3) / 4) * 4;
    std::vector<uint8_t> padded((size_t)wm->distinct * row + 16, 0);
    for (int j = 0; j < wm->distinct; ++j) memcpy(padded.data() + (size_t)j * row, wm->pat_sorted + (size_t)j * wm->m, (size_t)wm->m);
    P.pat_sorted = padded.data();
    P.gram_g7 = wm->gram_kind == SMH_GRAM_PAIR ? (const uint8_t *)wm->gram_table + SMH_GRAM_BYTES : nullptr;
    P.gram_planes = wm->gram_planes;
    if (wm->gram_kind == SMH_GRAM_FLAT || wm->gram_kind == SMH_GRAM_FLAT_BIG) P.gram_jb = wm->gram_jb; /* 1: two bits per gram */
    const uint64_t nthreads = blocks * EMU_BLOCK_THREADS;
    uint64_t total = 0;
    /* staged verify as launch_gram (wm_kernels.inc) picks it */
    int stg = wm->m - 1 <= 16 ? 1 : wm->m - 1 <= 32 ? 2 : 0;
    /* pair form with few survivors per chunk: in-register verify (STG 5 / 6), as launch_gram */
    const bool pairlike = wm->gram_kind == SMH_GRAM_PAIR || wm->gram_kind == SMH_GRAM_OCT2;
    bool regv = pairlike && stg > 0 && SMH_REGV_WANTED(wm->gram_density * 4096.0);
    if (const char *tn = getenv("SMH_WM_TUNE")) {
        if (strstr(tn, "regv=0")) regv = false;
        if (strstr(tn, "regv=1")) regv = pairlike && stg > 0;
    }
    /* the DNA forms through the windows-from-L2 pipeline (STG 3 / 4) from SMH_L2_MIN_PER_CHUNK to SMH_L2_DNA_MAX_PER_CHUNK
     * survivors per chunk, as launch_gram (late round 6); SMH_WM_TUNE "l2=0|1" as there */
    const bool l2_dna = wm->gram_kind == SMH_GRAM_PAIR || wm->gram_kind == SMH_GRAM_OCT2 || wm->gram_kind == SMH_GRAM_OCT;
    bool l2p = l2_dna && stg > 0 && wm->gram_density * 4096.0 >= (pairlike ? SMH_L2_MIN_PER_CHUNK_REGV : SMH_L2_MIN_PER_CHUNK) && wm->gram_density * 4096.0 <= SMH_L2_DNA_MAX_PER_CHUNK;
    if (const char *tn = getenv("SMH_WM_TUNE")) {
        if (strstr(tn, "l2=0") || strstr(tn, "regv=") || strstr(tn, "hd=")) l2p = false; /* (the knobs that name another stage) */
        if (strstr(tn, "l2=1")) l2p = l2_dna && stg > 0;
    }
    if (l2p) stg += 2; /* 3 / 4 */
    else if (regv) stg += 4;
    for (uint64_t t = 0; t < nthreads; ++t) {
        const smh_chunk_sched S = smh_sched_static(t >> 6, nthreads >> 6);
#define GRAM_CALL(KIND, STG) (po ? smh_wm_gram_thread<KIND, true, STG>(t, S, text, n, wm->gram_table, P, nullptr, po) \
                                 : smh_wm_gram_thread<KIND, false, STG>(t, S, text, n, wm->gram_table, P, nullptr, po))
#define GRAM_STG(KIND) (stg == 1 ? GRAM_CALL(KIND, 1) : stg == 2 ? GRAM_CALL(KIND, 2) : stg == 3 ? GRAM_CALL(KIND, 3) : stg == 4 ? GRAM_CALL(KIND, 4) : GRAM_CALL(KIND, 0))
        if (wm->gram_kind == SMH_GRAM_PAIR)
            total += stg == 5 ? GRAM_CALL(1, 5) : stg == 6 ? GRAM_CALL(1, 6) : GRAM_STG(1);
        else if (wm->gram_kind == SMH_GRAM_OCT2)
            total += stg == 5 ? GRAM_CALL(5, 5) : stg == 6 ? GRAM_CALL(5, 6) : GRAM_STG(5);
        else if (wm->gram_kind == SMH_GRAM_OCT)
            total += GRAM_STG(3);
        else if (wm->gram_kind == SMH_GRAM_BYTE_BIG) /* the 143.9 KiB table: windows from L2 always (wm_kernels.inc launch_gram) */
            total += stg == 1 ? GRAM_CALL(8, 3) : GRAM_CALL(8, 4);
        else if (wm->gram_kind == SMH_GRAM_FLAT4_BIG) /* four-byte grams (late round 6) */
            total += stg == 1 ? GRAM_CALL(11, 3) : GRAM_CALL(11, 4);
        else if (wm->gram_kind == SMH_GRAM_FLAT_BIG && wm->gram_jb > 0)
            total += stg == 1 ? GRAM_CALL(10, 3) : GRAM_CALL(10, 4);
        else if (wm->gram_kind == SMH_GRAM_FLAT_BIG)
            total += stg == 1 ? GRAM_CALL(9, 3) : GRAM_CALL(9, 4);
        else if (wm->gram_kind == SMH_GRAM_FLAT || wm->gram_kind == SMH_GRAM_BYTE) {
            /* the byte forms: windows from L2 (STG 3 / 4) unless SMH_WM_TUNE says "l2=0", as launch_gram */
            const char *tn = getenv("SMH_WM_TUNE");
            const bool l2 = stg > 0 && !(tn && strstr(tn, "l2=0"));
            if (wm->gram_kind == SMH_GRAM_FLAT && wm->gram_jb > 0) /* two bits per gram: kernel KIND 7 */
                total += l2 ? (stg == 1 ? GRAM_CALL(7, 3) : GRAM_CALL(7, 4)) : GRAM_STG(7);
            else if (wm->gram_kind == SMH_GRAM_FLAT)
                total += l2 ? (stg == 1 ? GRAM_CALL(6, 3) : GRAM_CALL(6, 4)) : GRAM_STG(6);
            else
                total += l2 ? (stg == 1 ? GRAM_CALL(2, 3) : GRAM_CALL(2, 4)) : GRAM_STG(2);
        }
#undef GRAM_STG
#undef GRAM_CALL
    }
    return total;
}

template <bool HASHED, bool EXACT, bool POS = false>
static uint64_t wm_halo(const smh_wm *wm, const uint8_t *text, uint64_t n, uint64_t blocks, const smh_pos_out *po = nullptr)
{
    const int halo = wm->m - 1;
    if constexpr (HASHED && !EXACT) {
        if (wm->filter_le4 && halo <= 32) { /* as launch_halo in wm_kernels.hip */
#define EMU_BYTE_BLOCK(HCV, STGV)                                                                                   \
            switch (wm->filter_k) {                                                                                   \
            case 2: return wm_grid<true, false, HCV, 2, POS, STGV>(wm, text, n, blocks, po);                           \
            case 3: return wm_grid<true, false, HCV, 3, POS, STGV>(wm, text, n, blocks, po);                           \
            case 5: return wm_grid<true, false, HCV, 5, POS, STGV>(wm, text, n, blocks, po);                           \
            default: return wm_grid<true, false, HCV, 4, POS, STGV>(wm, text, n, blocks, po);                          \
            }
            if (g_emu_n_classes == 0) { /* single-length set: staged verify */
                if (halo <= 16) { EMU_BYTE_BLOCK(1, true) }
                EMU_BYTE_BLOCK(2, true)
            }
            if (halo <= 16) { EMU_BYTE_BLOCK(1, false) }
            EMU_BYTE_BLOCK(2, false)
#undef EMU_BYTE_BLOCK
        }
    }
    if (halo <= 16) return wm_grid<HASHED, EXACT, 1, 0, POS>(wm, text, n, blocks, po);
    if (halo <= 32) return wm_grid<HASHED, EXACT, 2, 0, POS>(wm, text, n, blocks, po);
    if (halo <= 64) return wm_grid<HASHED, EXACT, 4, 0, POS>(wm, text, n, blocks, po);
    return wm_grid<HASHED, EXACT, 0, 0, POS>(wm, text, n, blocks, po);
}

extern "C" uint64_t emu_wm_scan(const smh_wm *wm, const uint8_t *text_in, uint64_t n, int variant, uint32_t blocks)
{
    if (n < (uint64_t)wm->m) return 0;
    if (!blocks) blocks = 4;
    uint64_t result[2];
    for (int mode = 0; mode < 2; ++mode) {
        guarded g = guard_copy(text_in, n, mode);
        const uint8_t *text = g.text;
        uint64_t total = 0;
        if (variant == SMH_VARIANT_TABLE) {
            uint16_t *sh = (uint16_t *)malloc(wm->shiftsize * 2);
            for (uint32_t i = 0; i < wm->shiftsize; ++i) sh[i] = (uint16_t)wm->l_shift[i];
            const uint64_t nthreads = (uint64_t)blocks * 256;
            for (uint64_t t = 0; t < nthreads; ++t)
                total += smh_wm_table_thread<uint16_t>(t, nthreads, text, n, sh, wm->shiftsize, wm->l_bucket_off,
                                                       wm->l_bucket, wm->pat_orig, wm->m, 2);
            free(sh);
        } else if (wm->pair_table) {
            const uint64_t nthreads = (uint64_t)blocks * EMU_BLOCK_THREADS;
            for (uint64_t t = 0; t < nthreads; ++t)
                total += smh_wm_pair_thread<SMH_PREFETCH != 0>(t, smh_sched_static(t >> 6, nthreads >> 6), text, n, wm->m, wm->pair_table, wm->filter);
        } else if (wm->gram_kind != SMH_GRAM_NONE) {
            total = wm_gram_grid(wm, text, n, blocks, nullptr);
        } else if (wm->filter_hashed) {
            total = wm_halo<true, false>(wm, text, n, blocks);
        } else if (wm->filter_exact) {
            total = wm_halo<false, true>(wm, text, n, blocks);
        } else {
            total = wm_halo<false, false>(wm, text, n, blocks);
        }
        guard_free(g);
        result[mode] = total;
    }
    return result[0] == result[1] ? result[0] : ~0ull;
}

/* ------------------------------------------------------------------ positions */
extern "C" uint64_t emu_ac_positions(const smh_ac *ac, const uint8_t *text, uint64_t n, uint64_t *out, uint64_t capacity,
                                     uint32_t blocks)
{
    if (!blocks) blocks = 2;
    size_t dflen = (size_t)ac->m + 2 < 72 ? 72 : (size_t)ac->m + 2;
    uint32_t *df = (uint32_t *)malloc(dflen * 4);
    for (size_t i = 0; i < dflen; ++i) df[i] = (int)i <= ac->max_depth + 1 ? ac->depth_first[i] : ac->rows;
    smh_ac_verify_ctx V = {};
    V.text = text; V.n = n; V.m = ac->m; V.K = ac->scan_depth; V.sigma = ac->alphabet;
    emu_hv hv;
    hv.fill(ac, V, df);
    uint64_t cursor = 0;
    const uint64_t nthreads = (uint64_t)blocks * 256;
    for (uint64_t t = 0; t < nthreads; ++t) smh_ac_positions_thread(t, nthreads, V, out, capacity, &cursor);
    free(df);
    return cursor;
}

extern "C" uint64_t emu_wm_positions(const smh_wm *wm, const uint8_t *text, uint64_t n, uint64_t *out, uint64_t capacity,
                                     uint32_t blocks)
{
    if (!blocks) blocks = 2;
    uint16_t *sh = (uint16_t *)malloc(wm->shiftsize * 2);
    for (uint32_t i = 0; i < wm->shiftsize; ++i) sh[i] = (uint16_t)wm->l_shift[i];
    uint64_t cursor = 0;
    const uint64_t nthreads = (uint64_t)blocks * 256;
    for (uint64_t t = 0; t < nthreads; ++t)
        smh_wm_positions_thread<uint16_t>(t, nthreads, text, n, sh, wm->shiftsize, wm->l_bucket_off, wm->l_bucket,
                                          wm->pat_orig, wm->m, 2, out, capacity, &cursor);
    free(sh);
    return cursor;
}

/* ------------------------------------------------------------------ SH */
#include "sh_lane.h"

/* variant TABLE: the reversed-trie walk with the bmBc skip loop (sh_lane.h); TUNED: the engine the
 * handle compiled (the Wu-Manber or the automaton lane code above).  bmbc NULL = the valid table. */
extern "C" uint64_t emu_sh_scan(const smh_sh *sh, const uint8_t *text_in, uint64_t n, const int32_t *bmbc, int variant,
                                uint32_t blocks)
{
    if (n < (uint64_t)sh->m) return 0;
    if (variant == SMH_VARIANT_TUNED)
        return sh->wm ? emu_wm_scan(sh->wm, text_in, n, SMH_VARIANT_TUNED, blocks)
                      : emu_ac_scan(sh->ac, text_in, n, SMH_VARIANT_TUNED, blocks);
    if (!blocks) blocks = 3;
    uint64_t result[2];
    for (int mode = 0; mode < 2; ++mode) {
        guarded g = guard_copy(text_in, n, mode);
        const uint64_t nthreads = (uint64_t)blocks * 256;
        uint64_t total = 0;
        for (uint64_t t = 0; t < nthreads; ++t)
            total += smh_sh_table_thread<int32_t>(t, nthreads, g.text, n, sh->g_transition, sh->g_final,
                                                  bmbc ? bmbc : sh->valid_bmbc, sh->m, sh->alphabet);
        guard_free(g);
        result[mode] = total;
    }
    return result[0] == result[1] ? result[0] : ~0ull;
}

/* ------------------------------------------------------------------ SBOM */
#include "sbom_lane.h"

extern "C" uint64_t emu_sbom_scan(const smh_sbom *sb, const uint8_t *text_in, uint64_t n, int variant, uint32_t blocks)
{
    if (n < (uint64_t)sb->m) return 0;
    if (variant == SMH_VARIANT_TUNED)
        return sb->wm ? emu_wm_scan(sb->wm, text_in, n, SMH_VARIANT_TUNED, blocks)
                      : emu_ac_scan(sb->ac, text_in, n, SMH_VARIANT_TUNED, blocks);
    if (!blocks) blocks = 3;
    uint64_t result[2];
    for (int mode = 0; mode < 2; ++mode) {
        guarded g = guard_copy(text_in, n, mode);
        const uint64_t nthreads = (uint64_t)blocks * 256;
        uint64_t total = 0;
        for (uint64_t t = 0; t < nthreads; ++t)
            total += smh_sbom_table_thread(t, nthreads, g.text, n, sb->g_transition, sb->g_final_off, sb->g_final_ids,
                                           sb->patterns, sb->m, sb->alphabet);
        guard_free(g);
        result[mode] = total;
    }
    return result[0] == result[1] ? result[0] : ~0ull;
}

/* the tuned WM kernels in positions mode (pair filter, exact / hashed / direct block filters) */
extern "C" uint64_t emu_wm_positions_tuned(const smh_wm *wm, const uint8_t *text_in, uint64_t n, uint64_t *out,
                                           uint64_t capacity, uint32_t blocks)
{
    if (n < (uint64_t)wm->m) return 0;
    if (!blocks) blocks = 2;
    std::vector<uint8_t> padded(((n + 15) / 16) * 16 + 64 + 64, 0);
    uint8_t *text = padded.data() + 64; /* the pair kernel reads the 16 bytes in front of a chunk (never of chunk 0) */
    memcpy(text, text_in, n);
    uint64_t cursor = 0;
    smh_pos_out po{out, capacity, &cursor};
    if (wm->pair_table) {
        const uint64_t nthreads = (uint64_t)blocks * EMU_BLOCK_THREADS;
        for (uint64_t t = 0; t < nthreads; ++t)
            smh_wm_pair_thread<false, true>(t, smh_sched_static(t >> 6, nthreads >> 6), text, n, wm->m, wm->pair_table, wm->filter, &po);
    } else if (wm->gram_kind != SMH_GRAM_NONE) {
        wm_gram_grid(wm, text, n, blocks, &po);
    } else if (wm->filter_hashed) {
        wm_halo<true, false, true>(wm, text, n, blocks, &po);
    } else if (wm->filter_exact) {
        wm_halo<false, true, true>(wm, text, n, blocks, &po);
    } else {
        wm_halo<false, false, true>(wm, text, n, blocks, &po);
    }
    return cursor;
}

/* a mixed-length set in one pass, as smh_wm_scan_multi / smh_wm_positions_multi run it: `suffix` is the
 * handle over the patterns' last min-length symbols, `classes` the per-length handles.  out == NULL:
 * count; else positions mode (returns the cursor). */
extern "C" uint64_t emu_wm_scan_multi(const smh_wm *suffix, const smh_wm *const *classes, int n_classes,
                                      const uint8_t *text_in, uint64_t n, uint64_t *out, uint64_t capacity, uint32_t blocks)
{
    if (n < (uint64_t)suffix->m) return 0;
    if (!blocks) blocks = 2;
    std::vector<uint8_t> padded(((n + 15) / 16) * 16 + 64 + 64, 0);
    uint8_t *text = padded.data() + 64;
    memcpy(text, text_in, n);
    std::vector<std::vector<uint8_t>> rows(n_classes);
    std::vector<smh_wm_class> cls(n_classes);
    for (int c = 0; c < n_classes; ++c) {
        const smh_wm *k = classes[c];
        const size_t row = (size_t)((k->m + 3) / 4) * 4;
        rows[c].assign((size_t)k->distinct * row + 16, 0);
        for (int j = 0; j < k->distinct; ++j) memcpy(rows[c].data() + (size_t)j * row, k->pat_sorted + (size_t)j * k->m, (size_t)k->m);
        cls[c].m = k->m;
        cls[c].verify_log2 = k->verify_log2;
        cls[c].verify = k->verify;
        cls[c].pat_sorted = rows[c].data();
    }
    g_emu_classes = cls.data();
    g_emu_n_classes = n_classes;
    uint64_t cursor = 0, total;
    smh_pos_out po{out, capacity, &cursor};
    if (suffix->gram_kind == SMH_GRAM_PAIR2) { /* grouped pair-gram filter over the full patterns */
        smh_wm_params P = {};
        P.m = suffix->m;
        P.bits = suffix->bits_per_symbol;
        P.n_classes = n_classes;
        P.classes = cls.data();
        P.gram_g7 = (const uint8_t *)suffix->gram_table + SMH_GRAM_BYTES;
        P.gram_jb = suffix->gram_jb;
        if (suffix->sfx_slot_off && !(getenv("SMH_WM_TUNE") && strstr(getenv("SMH_WM_TUNE"), "sfx=0"))) { /* the verify stage's suffix index, as the launcher sets it */
            const uint8_t *g = (const uint8_t *)suffix->gram_table;
            P.sfx_slot = (const uint32_t *)(g + suffix->sfx_slot_off);
            P.sfx_ent = (const uint32_t *)(g + suffix->sfx_ent_off);
            P.sfx_pat = (const uint32_t *)(g + suffix->sfx_pat_off);
        }
        const uint64_t nthreads = (uint64_t)blocks * EMU_BLOCK_THREADS;
        total = 0;
        for (uint64_t t = 0; t < nthreads; ++t) {
            const smh_chunk_sched S = smh_sched_static(t >> 6, nthreads >> 6);
            total += out ? smh_wm_gram_thread<4, true, 0>(t, S, text, n, suffix->gram_table, P, nullptr, &po)
                         : smh_wm_gram_thread<4, false, 0>(t, S, text, n, suffix->gram_table, P, nullptr, &po);
        }
        if (out) total = cursor;
        g_emu_classes = nullptr;
        g_emu_n_classes = 0;
        return total;
    }
    if (out) {
        total = suffix->filter_hashed ? wm_halo<true, false, true>(suffix, text, n, blocks, &po)
                                      : wm_halo<false, false, true>(suffix, text, n, blocks, &po);
        total = cursor;
    } else {
        total = suffix->filter_hashed ? wm_halo<true, false, false>(suffix, text, n, blocks)
                                      : wm_halo<false, false, false>(suffix, text, n, blocks);
    }
    g_emu_classes = nullptr;
    g_emu_n_classes = 0;
    return total;
}

/* ------------------------------------------------------------------ mixed-length automaton (acm_host.c, acm_lane.h) */
#include "acm_lane.h"

extern "C" uint64_t emu_acm_scan(const smh_acm *a, const uint8_t *text_in, uint64_t n, uint32_t blocks)
{
    if (!blocks) blocks = 3;
    uint64_t result[2];
    for (int mode = 0; mode < 2; ++mode) {
        guarded g = guard_copy(text_in, n, mode);
        smh_acm_ctx C;
        C.text = g.text; C.n = n; C.K = a->K; C.max_len = a->max_len; C.sigma = a->alphabet;
        C.g_goto = a->g_goto; C.g_final = a->g_final;
        const uint64_t nthreads = (uint64_t)blocks * EMU_BLOCK_THREADS;
        uint64_t total = 0;
        for (uint64_t t = 0; t < nthreads; ++t) {
            const smh_chunk_sched S = smh_sched_static(t >> 6, nthreads >> 6);
            if (a->entry_bytes == 2)
                total += a->alphabet == 4 ? smh_acm_thread<uint16_t, 4>(t, S, a->scan, a->scan, C, nullptr)
                                          : smh_acm_thread<uint16_t, 0>(t, S, a->scan, a->scan, C, nullptr);
            else
                total += a->alphabet == 4 ? smh_acm_thread<uint32_t, 4>(t, S, a->scan, a->scan, C, nullptr)
                                          : smh_acm_thread<uint32_t, 0>(t, S, a->scan, a->scan, C, nullptr);
        }
        guard_free(g);
        result[mode] = total;
    }
    return result[0] == result[1] ? result[0] : ~0ull;
}

/* ------------------------------------------------------------------ SOG table walk (sog_lane.h) */
#include "sog_lane.h"

extern "C" uint64_t emu_sog_scan(const smh_sog *sg, const uint8_t *text_in, uint64_t n, uint32_t blocks)
{
    if (n < 8) return 0;
    if (!blocks) blocks = 3;
    uint64_t result[2];
    for (int mode = 0; mode < 2; ++mode) {
        guarded g = guard_copy(text_in, n, mode);
        const uint64_t nthreads = (uint64_t)blocks * 256;
        uint64_t total = 0;
        for (uint64_t t = 0; t < nthreads; ++t)
            total += smh_sog_table_thread(t, nthreads, g.text, n, sg->t8, sg->hs, sg->index, sg->hs2, sg->patterns, (int)sg->n_patterns);
        guard_free(g);
        result[mode] = total;
    }
    return result[0] == result[1] ? result[0] : ~0ull;
}

/* ------------------------------------------------------------------ key engine (csrc/key_lane.h) */
#include "key_lane.h"
template <int KC, int HP, bool FULL>
static uint64_t keys_grid(const smh_keys *k, const uint8_t *text, uint64_t n, uint64_t blocks, uint64_t *out, uint64_t capacity, uint64_t *cursor)
{
    const uint64_t nthreads = blocks * EMU_BLOCK_THREADS;
    uint64_t total = 0;
    smh_pos_out po{out, capacity, cursor};
    for (uint64_t t = 0; t < nthreads; ++t) {
        const smh_chunk_sched S = smh_sched_static(t >> 6, nthreads >> 6);
        if (cursor) smh_key_thread<KC, HP, true, false>(t, S, text, n, k->image, k->P, &po);
        else total += smh_key_thread<KC, HP, false, FULL>(t, S, text, n, k->image, k->P, nullptr);
    }
    return cursor ? *cursor : total;
}
template <int R, int HP>
static uint64_t keyb_grid(const smh_keys *k, const uint8_t *text, uint64_t n, uint64_t blocks, uint64_t *out, uint64_t capacity, uint64_t *cursor)
{
    const uint64_t nthreads = blocks * EMU_BLOCK_THREADS;
    uint64_t total = 0;
    smh_pos_out po{out, capacity, cursor};
    for (uint64_t t = 0; t < nthreads; ++t) {
        const smh_chunk_sched S = smh_sched_static(t >> 6, nthreads >> 6);
        if (cursor) smh_keyb_thread<R, HP, true>(t, S, text, n, k->image, k->P, &po);
        else total += smh_keyb_thread<R, HP, false>(t, S, text, n, k->image, k->P, nullptr);
    }
    return cursor ? *cursor : total;
}
static uint64_t keys_any(const smh_keys *k, const uint8_t *text, uint64_t n, uint64_t blocks, uint64_t *out, uint64_t capacity, uint64_t *cursor)
{
    if (k->P.layout == 1) { /* the bucket image: as key_kernels.hip launch_any */
        if (k->P.bk_old == 15) return keyb_grid<15, 2>(k, text, n, blocks, out, capacity, cursor);
        if (k->P.bk_old == 6) return keyb_grid<6, 1>(k, text, n, blocks, out, capacity, cursor);
        return k->P.m - 1 > 16 ? keyb_grid<0, 2>(k, text, n, blocks, out, capacity, cursor) : keyb_grid<0, 1>(k, text, n, blocks, out, capacity, cursor);
    }
    const bool hp2 = k->P.m - 1 > 16;
    const int kb = k->P.m * k->P.bits; /* as key_kernels.hip launch_any: keys that fill their slot run the instantiation without the mask */
    if (kb == 64) return hp2 ? keys_grid<1, 2, true>(k, text, n, blocks, out, capacity, cursor) : keys_grid<1, 1, true>(k, text, n, blocks, out, capacity, cursor);
    if (kb == 32) return hp2 ? keys_grid<0, 2, true>(k, text, n, blocks, out, capacity, cursor) : keys_grid<0, 1, true>(k, text, n, blocks, out, capacity, cursor);
    if (k->P.wide == 2) return hp2 ? keys_grid<2, 2, false>(k, text, n, blocks, out, capacity, cursor) : keys_grid<2, 1, false>(k, text, n, blocks, out, capacity, cursor);
    if (k->P.wide == 1) return hp2 ? keys_grid<1, 2, false>(k, text, n, blocks, out, capacity, cursor) : keys_grid<1, 1, false>(k, text, n, blocks, out, capacity, cursor);
    return hp2 ? keys_grid<0, 2, false>(k, text, n, blocks, out, capacity, cursor) : keys_grid<0, 1, false>(k, text, n, blocks, out, capacity, cursor);
}
/* count; the text once ending at and once starting behind a guard page (both results must agree: ~0 if not) */
extern "C" uint64_t emu_keys_scan(const smh_keys *k, const uint8_t *text_in, uint64_t n, uint32_t blocks)
{
    if (n < (uint64_t)k->m) return 0;
    if (!blocks) blocks = 3;
    uint64_t result[2];
    for (int mode = 0; mode < 2; ++mode) {
        guarded g = guard_copy(text_in, n, mode);
        result[mode] = keys_any(k, g.text, n, blocks, nullptr, 0, nullptr);
        guard_free(g);
    }
    return result[0] == result[1] ? result[0] : ~0ull;
}
extern "C" uint64_t emu_keys_positions(const smh_keys *k, const uint8_t *text_in, uint64_t n, uint64_t *out, uint64_t capacity, uint32_t blocks)
{
    if (n < (uint64_t)k->m) return 0;
    if (!blocks) blocks = 2;
    guarded g = guard_copy(text_in, n, 0);
    uint64_t cursor = 0;
    keys_any(k, g.text, n, blocks, out, capacity, &cursor);
    guard_free(g);
    return cursor;
}

/* ------------------------------------------------------------------ window-hash engine (csrc/hash_lane.h) */
#include "hash_lane.h"
/* count (out == NULL) or positions of a Wu-Manber handle's window-hash engine; ~0 = the handle keeps none */
extern "C" uint64_t emu_hash_scan(const smh_wm *wm, const uint8_t *text_in, uint64_t n, uint64_t *out, uint64_t capacity, uint32_t blocks,
                                  uint64_t *events_out)
{
    const smh_hashes *k = wm->hashes;
    if (!k) return ~0ull;
    if (n < (uint64_t)k->m) return 0;
    if (!blocks) blocks = 3;
    uint64_t result[2] = {0, 0}, cursor = 0, events = 0;
    for (int mode = 0; mode < (out ? 1 : 2); ++mode) {
        guarded g = guard_copy(text_in, n, mode);
        smh_hash_ctx C;
        C.text = g.text; C.n = n; C.P = k->P; C.table = k->table; C.drop = 0;
        smh_pos_out po{out, capacity, &cursor};
        const uint64_t nthreads = (uint64_t)blocks * EMU_BLOCK_THREADS;
        uint64_t total = 0;
        events = 0;
        for (uint64_t t = 0; t < nthreads; ++t) {
            const smh_chunk_sched S = smh_sched_static(t >> 6, nthreads >> 6);
            uint32_t ev = 0;
            switch ((k->m + 3) / 4) { /* as hash_kernels.hip launch_nd */
#define EMU_HASH_ND(ND) case ND: if (out) smh_hash_thread<true, ND>(t, S, C, k->bloom, nullptr, &po, &ev); else if (k->P.bloom_k >= 3u) total += smh_hash_thread<false, ND, true>(t, S, C, k->bloom, nullptr, nullptr, &ev); else total += smh_hash_thread<false, ND>(t, S, C, k->bloom, nullptr, nullptr, &ev); break;
            EMU_HASH_ND(1) EMU_HASH_ND(2) EMU_HASH_ND(3) EMU_HASH_ND(4) EMU_HASH_ND(5) EMU_HASH_ND(6) EMU_HASH_ND(7)
            default: if (out) smh_hash_thread<true, 8>(t, S, C, k->bloom, nullptr, &po, &ev); else if (k->P.bloom_k >= 3u) total += smh_hash_thread<false, 8, true>(t, S, C, k->bloom, nullptr, nullptr, &ev); else total += smh_hash_thread<false, 8>(t, S, C, k->bloom, nullptr, nullptr, &ev); break;
#undef EMU_HASH_ND
            }
            events += ev;
        }
        result[mode] = total;
        guard_free(g);
    }
    if (events_out) *events_out = events;
    if (out) return cursor;
    return result[0] == result[1] ? result[0] : ~0ull - 1;
}
