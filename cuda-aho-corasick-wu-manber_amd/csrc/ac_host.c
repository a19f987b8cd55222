/*
 * csrc/ac_host.c -- host side of the Aho-Corasick path (plain C).
 *
 *   preproc_ac / free_ac      drop-in for ac/ac.c:224-252: fills the caller's
 *                             state_transition / state_supply / state_final
 *                             exactly as the reference does.
 *   smh_ac_compile_tables     goto/supply/final tables -> complete DFA laid out
 *                             for the gfx950 kernels (smh_internal.h).
 *
 * Design notes (how this differs from the reference, not what it computes):
 *   - the trie lives directly in the caller's flat state_transition table
 *     (-1 = no edge; row 0: 0 = no edge), no node structs or per-node malloc
 *     (reference: ac/ac.c:148-171);
 *   - failure links come from an array-queue BFS, O(states * alphabet)
 *     (reference: list_append walks the whole queue, ac/list.h:57-74 -> quadratic);
 *   - search never runs on the host: search_ac (smh_runtime.hip) launches the kernel.
 */
#include "smh_internal.h"
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ errors */
static __thread char g_err[512];

void smh_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

const char *smh_last_error(void) { return g_err; }
_Thread_local int smh_alt_engine_depth = 0;
const char *smh_version(void) { return "mi355x-smatcher 0.3 (gfx950)"; }

uint64_t smh_handle_serial(void)
{
    static uint64_t next = 0;
    return __atomic_add_fetch(&next, 1, __ATOMIC_RELAXED);
}

/* reference: fail() from the absent ../helper2.h -- message, then exit */
void fail(const char *msg)
{
    fputs(msg, stderr);
    exit(1);
}

/* ------------------------------------------------------------------ preproc_ac */
static inline int has_edge(const int *trans, uint32_t state, int32_t v)
{
    (void)trans;
    return state == 0 ? v > 0 : v != -1;
}

/* the reference's table construction (ac_init + ac_addstring + ac_maketree) on the caller's flat
 * arrays; returns the number of states in *idcounter_out */
static void ac_fill_tables(unsigned char **pattern, int m, int p_size, int alphabet,
                           int *state_transition, unsigned int *state_supply,
                           unsigned int *state_final, uint32_t *idcounter_out,
                           uint32_t *patterncounter_out)
{
    if (m < 1 || p_size < 0 || alphabet < 1 || alphabet > 256)
        fail("preproc_ac: bad arguments\n");
    const size_t A = (size_t)alphabet;
    /* ac_init, ac/ac.c:59-62: row 0 of the flat table is all zero */
    for (int c = 0; c < alphabet; ++c) state_transition[c] = 0;
    uint32_t idcounter = 1, patterncounter = 0;

    /* ac_addstring, ac/ac.c:127-196: new state id = creation order */
    for (int j = 0; j < p_size; ++j) {
        const unsigned char *s = pattern[j];
        uint32_t state = 0;
        for (int i = 0; i < m; ++i) {
            unsigned c = s[i];
            if ((int)c >= alphabet) fail("preproc_ac: pattern symbol outside the alphabet\n");
            int32_t nx = state_transition[state * A + c];
            if (!has_edge(state_transition, state, nx)) {
                nx = (int32_t)idcounter++;
                state_transition[state * A + c] = nx;
            }
            state = (uint32_t)nx;
        }
        if (!state_final[state]) {
            state_final[state] = 1;
            ++patterncounter;
        }
    }

    /* ac_maketree, ac/ac.c:79-124: breadth-first failure links; like the reference they are
     * kept privately (struct ac_state.fail there, failv[] here) and exported to
     * state_supply[] for depth >= 2 states only (ac/ac.c:114) */
    uint32_t *queue = (uint32_t *)malloc((size_t)idcounter * sizeof(uint32_t));
    uint32_t *failv = (uint32_t *)calloc((size_t)idcounter, sizeof(uint32_t));
    if (!queue || !failv) fail("Could not allocate memory\n");
    size_t head = 0, tail = 0;
    for (int c = 0; c < alphabet; ++c)
        if (state_transition[c] > 0) queue[tail++] = (uint32_t)state_transition[c];
    while (head < tail) {
        uint32_t cur = queue[head++];
        for (int c = 0; c < alphabet; ++c) {
            int32_t s = state_transition[cur * A + c];
            if (s == -1) continue;
            queue[tail++] = (uint32_t)s;
            uint32_t st = failv[cur];
            int32_t t;
            for (;;) {
                t = state_transition[st * A + c];
                if (st == 0 || t != -1) break;
                st = failv[st];
            }
            /* at the root a missing edge reads 0 == the root itself (ac/ac.c:86-88) */
            failv[s] = (uint32_t)t;
            state_supply[s] = (uint32_t)t;
        }
    }
    free(queue);
    free(failv);
    *idcounter_out = idcounter;
    *patterncounter_out = patterncounter;
}

struct ac_table *preproc_ac(unsigned char **pattern, int m, int p_size, int alphabet,
                            int *state_transition, unsigned int *state_supply,
                            unsigned int *state_final)
{
    struct smh_ac_table_box *box = (struct smh_ac_table_box *)calloc(1, sizeof *box);
    if (!box) fail("Could not initialize table\n");
    uint32_t idcounter, patterncounter;
    ac_fill_tables(pattern, m, p_size, alphabet, state_transition, state_supply, state_final,
                   &idcounter, &patterncounter);
    const size_t A = (size_t)alphabet;

    box->pub.idcounter = idcounter;
    box->pub.patterncounter = patterncounter;
    box->root.id = 0;
    box->root.keywordline = 0;
    box->root.output = NULL;
    box->root.fail = &box->root;
    box->root.next = (struct ac_state **)malloc(A * sizeof(struct ac_state *));
    if (!box->root.next) fail("Could not allocate memory\n");
    for (int c = 0; c < alphabet; ++c) box->root.next[c] = &box->root;
    box->pub.zerostate = &box->root;
    box->magic = SMH_MAGIC_AC;
    /* search_ac only runs the tuned kernel: no reference-layout copy for this handle */
    box->ac = smh_ac_compile_tables_impl(state_transition, state_supply, state_final,
                                         (uint64_t)idcounter, alphabet, m, SMH_AC_REF_NONE);
    if (!box->ac) {
        fputs(smh_last_error(), stderr);
        fail("\npreproc_ac: could not compile the automaton\n");
    }
    return &box->pub;
}

void free_ac(struct ac_table *table, int alphabet)
{
    (void)alphabet;
    if (!table) return;
    struct smh_ac_table_box *box = (struct smh_ac_table_box *)table;
    if (box->magic != SMH_MAGIC_AC) fail("free_ac: not a table from preproc_ac\n");
    smh_ac_free(box->ac);
    free(box->root.next);
    box->magic = 0;
    free(box);
}

/* ------------------------------------------------------------------ DFA compile */
void smh_ac_host_free(struct smh_ac *ac)
{
    if (!ac) return;
    free(ac->table);
    free(ac->row_depth);
    free(ac->row_fail);
    if (ac->trunc1_table != ac->scan_table) free(ac->trunc1_table);
    free(ac->scan_table);
    free(ac->depth_first);
    free(ac->g_transition);
    free(ac->g_supply);
    free(ac->g_final);
    free(ac->dense_pair);
    free(ac->dense_filter);
    if (ac->hv_wm != ac->alt_wm) smh_wm_free(ac->hv_wm); /* one handle may serve both roles */
    smh_wm_free(ac->alt_wm);
    smh_ac_free(ac->flat_ac);
    smh_ac_free(ac->flat_next);
    smh_keys_free(ac->keys);
    ac->magic = 0;
    free(ac);
}

/* the chosen automaton plan in ms per GiB on MI355X: the plan model's cost is in units of the exact stride-1 scan
 * (0.289 ms/GiB) and ranks the automaton plans among themselves; the hybrid image's kernels measure 0.183 (exact, two
 * chains + prefetch) and 0.205 (depth-cut, three chains) where the model says 0.15 / 0.17 (profiles/r03_q) */
double smh_ac_plan_ms(const struct smh_ac *ac)
{
    double ms = ac->scan_cost * 0.289;
    if (ac->scan_full_rows) ms += ac->scan_exact ? 0.03 : 0.035;
    return ms;
}

/* the patterns of a fixed-length goto trie, depth-first: `count` strings of m symbols (NULL on a
 * malformed trie or when memory runs out -- the caller then keeps the automaton engine) */
static unsigned char *ac_extract_patterns(const int *trans, const unsigned int *final, uint32_t R, int alphabet, int m,
                                          uint32_t count)
{
    const size_t A = (size_t)alphabet;
    unsigned char *out = (unsigned char *)malloc((size_t)(count ? count : 1) * (size_t)m);
    uint32_t *stack = (uint32_t *)malloc(((size_t)m + 2) * sizeof(uint32_t));
    int *sym = (int *)malloc(((size_t)m + 2) * sizeof(int));
    unsigned char *path = (unsigned char *)malloc((size_t)m + 1);
    size_t np = 0;
    int ok = out && stack && sym && path;
    if (ok) {
        int depth = 0;
        stack[0] = 0;
        sym[0] = 0;
        while (depth >= 0 && ok) {
            const uint32_t r = stack[depth];
            if (depth == m) {
                if (final[r]) {
                    if (np >= count) { ok = 0; break; }
                    memcpy(out + np * (size_t)m, path, (size_t)m);
                    ++np;
                }
                --depth;
                continue;
            }
            int c = sym[depth];
            int32_t s = -1;
            for (; c < alphabet; ++c) {
                s = trans[r * A + (size_t)c];
                if (has_edge(trans, r, s)) break;
            }
            if (c >= alphabet) { --depth; continue; }
            sym[depth] = c + 1;
            if ((uint32_t)s >= R) { ok = 0; break; }
            path[depth] = (unsigned char)c;
            ++depth;
            stack[depth] = (uint32_t)s;
            sym[depth] = 0;
        }
    }
    free(stack); free(sym); free(path);
    if (!ok || np != count) { free(out); return NULL; }
    return out;
}

struct smh_ac *smh_ac_compile_tables_impl(const int *trans, const unsigned int *supply,
                                          const unsigned int *final, uint64_t rows_in,
                                          int alphabet, int m, int ref_mode)
{
    if (!trans || !supply || !final || rows_in < 1 || alphabet < 1 || alphabet > 256 || m < 1) {
        smh_set_error("smh_ac_compile_tables: bad arguments");
        return NULL;
    }
    if (rows_in > 0x7FFFFFFFull) {
        smh_set_error("smh_ac_compile_tables: more than 2^31 states");
        return NULL;
    }
    const size_t A = (size_t)alphabet;
    const uint32_t R = (uint32_t)rows_in;

    /* 1. breadth-first sweep over goto edges: order, depth, leaf flag */
    uint32_t *order = (uint32_t *)malloc((size_t)R * sizeof(uint32_t));
    uint32_t *depth = (uint32_t *)malloc((size_t)R * sizeof(uint32_t));
    uint8_t *seen = (uint8_t *)calloc(R, 1);
    uint8_t *leaf = (uint8_t *)malloc(R);
    uint32_t *canon = (uint32_t *)malloc((size_t)R * sizeof(uint32_t));
    uint32_t *newid = (uint32_t *)malloc((size_t)R * sizeof(uint32_t));
    struct smh_ac *ac = (struct smh_ac *)calloc(1, sizeof *ac);
    if (ac) ac->engine_forced = -1, ac->serial = smh_handle_serial();
    uint32_t *full = NULL;
    if (!order || !depth || !seen || !leaf || !canon || !newid || !ac) goto oom;

    size_t head = 0, tail = 0;
    order[tail++] = 0;
    depth[0] = 0;
    seen[0] = 1;
    uint32_t max_id = 0;
    while (head < tail) {
        uint32_t cur = order[head++];
        int kids = 0;
        for (int c = 0; c < alphabet; ++c) {
            int32_t s = trans[cur * A + c];
            if (!has_edge(trans, cur, s)) continue;
            if ((uint32_t)s >= R || seen[s]) {
                smh_set_error("smh_ac_compile_tables: state_transition is not a trie "
                              "(edge %u -> %d)", cur, s);
                goto bad;
            }
            seen[s] = 1;
            depth[s] = depth[cur] + 1;
            order[tail++] = (uint32_t)s;
            if ((uint32_t)s > max_id) max_id = (uint32_t)s;
            ++kids;
        }
        leaf[cur] = kids == 0;
    }
    const uint32_t nstates = (uint32_t)tail;

    /* 2. fold leaves onto their supply chain; number the kept states breadth-first */
    uint32_t rows = 0, finals = 0;
    int fixed_ok = 1;
    for (uint32_t k = 0; k < nstates; ++k) {
        uint32_t u = order[k];
        uint32_t sup = depth[u] <= 1 ? 0u : supply[u];
        if (u != 0 && (sup >= R || !seen[sup] || depth[sup] >= depth[u])) {
            smh_set_error("smh_ac_compile_tables: bad supply link %u -> %u", u, sup);
            goto bad;
        }
        if (final[u]) {
            ++finals;
            if (!leaf[u] || depth[u] != (uint32_t)m) fixed_ok = 0;
        } else if (leaf[u] && u != 0) {
            fixed_ok = 0; /* a non-accepting leaf cannot come from preproc_ac */
        }
        if (u == 0 || !leaf[u]) {
            canon[u] = u;
            newid[u] = rows++;
        } else {
            canon[u] = canon[sup];
            newid[u] = 0xFFFFFFFFu;
        }
    }

    /* 3. complete the transition function over kept states, breadth-first:
     *    delta(s,c) = goto(s,c) if defined, else delta(supply(s),c)   (ac/ac.c:209-211) */
    full = (uint32_t *)malloc((size_t)rows * A * sizeof(uint32_t));
    if (!full) goto oom;
    for (uint32_t k = 0; k < nstates; ++k) {
        uint32_t u = order[k];
        if (canon[u] != u) continue;
        uint32_t *row = full + (size_t)newid[u] * A;
        if (u == 0) {
            for (int c = 0; c < alphabet; ++c) {
                int32_t s = trans[c];
                row[c] = has_edge(trans, 0, s) ? (uint32_t)s : 0u;
            }
        } else {
            uint32_t sup = depth[u] <= 1 ? 0u : supply[u];
            const uint32_t *srow = full + (size_t)newid[canon[sup]] * A;
            for (int c = 0; c < alphabet; ++c) {
                int32_t s = trans[u * A + c];
                row[c] = s != -1 ? (uint32_t)s : srow[c];
            }
        }
    }

    /* 4. encode: entry = row(canon(target)) | FLAG(final[target]) */
    ac->magic = SMH_MAGIC_AC;
    ac->alphabet = alphabet;
    ac->m = m;
    ac->states = nstates;
    ac->finals = finals;
    ac->rows = rows;
    ac->fixed_length_ok = fixed_ok;
    ac->entry_bytes = rows <= 32768u ? 2 : 4;
    ac->table_bytes = (uint64_t)rows * A * (uint64_t)ac->entry_bytes;
    {
        /* one lookup per entry (the encoded value of every reference state, computed once), and in
         * place: `full` becomes the table -- first-touch page faults on another rows*alphabet array
         * cost more than the arithmetic for alphabet-256 automata */
        const uint32_t flag = ac->entry_bytes == 2 ? 0x8000u : 0x80000000u;
        uint32_t *enc = (uint32_t *)malloc((size_t)R * sizeof(uint32_t));
        if (!enc) goto oom;
        for (uint32_t k = 0; k < nstates; ++k) {
            const uint32_t t = order[k];
            enc[t] = newid[canon[t]] | (final[t] ? flag : 0u);
        }
        const size_t total = (size_t)rows * A;
        if (ac->entry_bytes == 2) {
            uint16_t *dst = (uint16_t *)full; /* front to back: entry i is read before it is overwritten */
            for (size_t i = 0; i < total; ++i) dst[i] = (uint16_t)enc[full[i]];
            ac->table = realloc(full, total ? total * 2 : 2);
        } else {
            for (size_t i = 0; i < total; ++i) full[i] = enc[full[i]];
            ac->table = full;
        }
        full = NULL;
        free(enc);
        if (!ac->table) goto oom;
    }

    /* 5. depth_first[d] = first row whose depth >= d (rows are in BFS order) */
    int max_depth = 0;
    for (uint32_t k = 0; k < nstates; ++k)
        if (canon[order[k]] == order[k] && (int)depth[order[k]] > max_depth)
            max_depth = (int)depth[order[k]];
    ac->max_depth = max_depth;
    ac->depth_first = (uint32_t *)malloc((size_t)(max_depth + 2) * sizeof(uint32_t));
    if (!ac->depth_first) goto oom;
    {
        int d = 0;
        uint32_t r = 0;
        for (uint32_t k = 0; k < nstates; ++k) {
            uint32_t u = order[k];
            if (canon[u] != u) continue;
            while (d <= (int)depth[u]) ac->depth_first[d++] = r;
            ++r;
        }
        while (d <= max_depth + 1) ac->depth_first[d++] = rows;
    }

    ac->row_depth = (uint8_t *)malloc(rows ? rows : 1);
    ac->row_fail = (uint32_t *)malloc((size_t)(rows ? rows : 1) * sizeof(uint32_t));
    if (!ac->row_depth || !ac->row_fail) goto oom;
    for (uint32_t k = 0; k < nstates; ++k) {
        uint32_t u = order[k];
        if (canon[u] != u) continue;
        uint32_t sup = depth[u] <= 1 ? 0u : supply[u];
        ac->row_depth[newid[u]] = (uint8_t)(depth[u] > 255 ? 255 : depth[u]);
        ac->row_fail[newid[u]] = newid[canon[sup]];
    }

    /* 6. reference-layout tables, truncated to the ids in use, for SMH_VARIANT_TABLE: copied
     *    (caller keeps its arrays), adopted (malloc'ed arrays handed over, patched in place -- a
     *    second states*alphabet array is the most expensive thing in an alphabet-256 compile), or none */
    if (ref_mode != SMH_AC_REF_NONE) {
        size_t keep = (size_t)max_id + 1;
        if (ref_mode == SMH_AC_REF_COPY) {
            ac->g_transition = (int32_t *)malloc(keep * A * sizeof(int32_t));
            ac->g_supply = (uint32_t *)malloc(keep * sizeof(uint32_t));
            ac->g_final = (uint32_t *)malloc(keep * sizeof(uint32_t));
            if (!ac->g_transition || !ac->g_supply || !ac->g_final) goto oom;
            memcpy(ac->g_transition, trans, keep * A * sizeof(int32_t));
        } else {
            /* patched in place; the block is shrunk to the ids in use only once the compile can no longer
             * fail (below), so that on every failure path the caller still owns exactly what it passed in */
            ac->g_transition = (int32_t *)trans;
            ac->g_supply = (uint32_t *)supply;
            ac->g_final = (uint32_t *)final;
        }
        for (size_t u = 0; u < keep; ++u) {
            /* depth <= 1 states: the reference never writes their supply entry; the walk needs 0 */
            ac->g_supply[u] = (seen[u] && depth[u] >= 2) ? supply[u] : 0u;
            ac->g_final[u] = seen[u] ? (final[u] ? 1u : 0u) : 0u;
        }
        ac->states = (uint32_t)keep; /* == idcounter when the tables came from preproc_ac */
    }

    free(order); free(depth); free(seen); free(leaf); free(canon); free(newid); free(full);
    order = depth = canon = newid = full = NULL;
    seen = leaf = NULL;
    if (ac->fixed_length_ok && smh_ac_plan_scan(ac, SMH_AC_LDS_BUDGET, 0, 0) != SMH_OK) goto bad;
    /* Scan-engine choice.  When even the best LDS automaton lets so many candidates through that the
     * verify stage dominates (alphabet-256 sets: K = 1; thousands of long DNA patterns: K = 8, one
     * position in nine) the same count is obtained much faster by the suffix-filter kernels, which
     * hash W symbols instead of walking K levels.  The patterns are read back from the goto trie and
     * compiled for that engine; smh_ac_scan / smh_ac_positions use it unless a plan is forced. */
    /* The same Wu-Manber handle also serves the automaton kernels' own verify stage (`hv_wm`): a candidate of a depth-cut
     * plan (K < m) used to be walked down the full DFA -- up to m DEPENDENT loads, ~1 us each beside the streaming text,
     * and every wave ends with such a walk: 20-30 us at the end of a 200 us launch (tools/wavetrace.py).  Hashing the
     * window and probing the handle's verify table decides it in three. */
    /* Only plans that need it pay for it: an exact plan (K == m) never verifies a candidate and a cheap plan never hands
     * the scan over, so for those the patterns are not extracted and no second handle is compiled (it doubled the compile
     * time and the host memory of the common exact plans).  A depth-cut plan forced later on such a handle
     * (smh_ac_set_scan_plan) walks its candidates down the full DFA instead -- same count. */
    if (ac->fixed_length_ok && m >= 3 && smh_wu_shiftsize_for(alphabet) && smh_alt_engine_depth == 0 &&
        !ac->scan_dense && (!ac->scan_exact || ac->scan_cost > SMH_AC_ALT_ENGINE_COST)) {
        ++smh_alt_engine_depth;
        /* the caller's arrays may have been adopted (and shrunk in place) in step 6: read the handle's copy */
        const int *tsrc = ac->g_transition ? ac->g_transition : trans;
        const unsigned int *fsrc = ac->g_final ? ac->g_final : final;
        unsigned char *pats = ac_extract_patterns(tsrc, fsrc, ac->g_transition ? ac->states : R, alphabet, m, ac->finals);
        if (pats) {
            struct smh_wm *w = smh_wm_compile(pats, m, (int)ac->finals, alphabet); /* NULL: stay with the automaton / the walk */
            free(pats);
            /* Round 3: a depth-cut plan is a prefix filter with a verify stage behind it, and so is the pair-gram
             * shift-or filter -- but that one covers the WHOLE pattern with its planes (next to nothing survives) and
             * its lookups do not depend on each other, where the automaton's form one chain per lane: measured on the
             * headline sets (1000 patterns of 16 / 32 symbols) 0.174 against 0.205 ms/GiB.  The faster estimate scans;
             * the handle keeps serving the automaton kernels' verify stage when a plan is forced. */
            const int verify_bound = ac->scan_cost > SMH_AC_ALT_ENGINE_COST;
            const int filter_faster = w && !ac->scan_exact && w->gram_kind != SMH_GRAM_NONE && !w->alt_ac &&
                                      w->scan_ms_est + SMH_AC_ALT_ENGINE_MARGIN_MS < smh_ac_plan_ms(ac);
            if (w && (verify_bound || filter_faster)) ac->alt_wm = w;
            if (w && !verify_bound) ac->hv_wm = w;
            /* both engines stay at hand when both could serve (round 4): which one runs is then decided per text */
            if (w && !verify_bound && !ac->scan_exact && !w->alt_ac && !w->pair_table && (w->gram_kind != SMH_GRAM_NONE || !w->filter_exact) &&
                w->scan_ms_est > 0)
                ac->flex_wm = w;
        }
        --smh_alt_engine_depth;
    }
    if (ref_mode == SMH_AC_REF_ADOPT && ac->g_transition) {
        /* nothing can fail any more: ownership has passed, shrink the adopted block (a failed shrink leaves it valid) */
        void *t = realloc(ac->g_transition, (size_t)ac->states * (size_t)alphabet * sizeof(int32_t));
        if (t) ac->g_transition = (int32_t *)t;
    }
    /* Round 4: the hybrid image is fast while the lanes stay in its full rows -- on random text nearly always.  On text
     * that keeps matching long pattern prefixes (repeats, low-complexity runs, the patterns themselves recurring: the
     * reference's genomes and proteins, main.c:39-109) most lanes sit in compact rows and every step takes the resolution
     * path: measured 2-4.5 ms/GiB where the plan says 0.2.  A PLAIN stride-1 image has no such path: one lookup per symbol
     * whatever the text, 0.29 ms/GiB.  When the whole automaton fits LDS in that form (K = m: exact, no verify stage) it is
     * kept beside the preferred plan as the engine with a guarantee, and the runtime switches to it when the launches
     * report that the text is of that kind (smh_runtime.hip "adaptive engine"). */
    /* The same guarantee for a set whose automaton does NOT fit LDS whole: the patterns in trie order cut into the fewest
     * runs whose own automata do (a run's rows = 1 + the symbols its patterns do not share with their predecessor), one
     * exact stride-1 image per run, scanned one after the other over the same text -- P launches of 0.27-0.29 ms/GiB
     * whatever the text, where 1000 patterns of 32 symbols on repeat-rich DNA measured 2.4 ms/GiB through the filter
     * kernels and 6.0 through the hybrid image (profiles/r04_final/bench.json "skewed").  Up to SMH_FLAT_MAX_PARTS runs;
     * built from the patterns read back from the goto trie, so that a handle from preproc_ac has it too. */
    static _Thread_local int flat_building = 0; /* the parts are compiled by this very function */
    const uint64_t flat_cap_rows = SMH_AC_LDS_BUDGET / ((uint64_t)alphabet * 2u) < 32768u ? SMH_AC_LDS_BUDGET / ((uint64_t)alphabet * 2u) : 32768u;
    if (ac->fixed_length_ok && !ac->scan_dense && !(ac->scan_exact && !ac->scan_full_rows) && !flat_building && smh_alt_engine_depth <= 1 &&
        flat_cap_rows > (uint64_t)m + 1 && (uint64_t)ac->rows <= flat_cap_rows * SMH_FLAT_MAX_PARTS) {
        ++flat_building;
        ++smh_alt_engine_depth;
        const int *tsrc = ac->g_transition ? ac->g_transition : trans;
        const unsigned int *fsrc = ac->g_final ? ac->g_final : final;
        unsigned char *pats = ac_extract_patterns(tsrc, fsrc, ac->g_transition ? ac->states : R, alphabet, m, ac->finals);
        struct smh_ac *head = NULL, **link = &head;
        int parts = 0, ok = pats != NULL;
        for (uint32_t j0 = 0; ok && j0 < ac->finals;) {
            /* the longest run from j0 whose trie has at most flat_cap_rows nodes */
            uint64_t rows_run = 1u + (uint64_t)m;
            uint32_t j1 = j0 + 1;
            for (; j1 < ac->finals; ++j1) {
                const unsigned char *a = pats + (size_t)(j1 - 1) * (size_t)m, *b = pats + (size_t)j1 * (size_t)m;
                int lcp = 0;
                while (lcp < m && a[lcp] == b[lcp]) ++lcp;
                if (rows_run + (uint64_t)(m - lcp) > flat_cap_rows) break;
                rows_run += (uint64_t)(m - lcp);
            }
            struct smh_ac *part = ++parts <= SMH_FLAT_MAX_PARTS ? smh_ac_compile_patterns(pats + (size_t)j0 * (size_t)m, m, (int)(j1 - j0), alphabet) : NULL;
            if (part && (!part->fixed_length_ok || smh_ac_plan_scan(part, SMH_AC_LDS_BUDGET, 1, 0) != SMH_OK || !part->scan_exact ||
                         part->scan_full_rows || part->scan_stride != 1 || part->scan_dense)) {
                smh_ac_free(part);
                part = NULL;
            }
            if (!part) { ok = 0; break; }
            *link = part;
            link = &part->flat_next;
            j0 = j1;
        }
        free(pats);
        if (!ok) { smh_ac_free(head); head = NULL; parts = 0; }
        ac->flat_ac = head;
        ac->flat_parts = head ? parts : 0;
        --smh_alt_engine_depth;
        --flat_building;
    }
    /* Round 5: the key engine (key_hash.h) -- the set as a hash set of its m-symbol windows in LDS, ONE exact pass whatever the
     * text -- for every handle whose own plan is text-dependent or verify-bound (the same handles that look for flat parts) and
     * that is not served by a single plain stride-1 image already (0.25 ms/GiB: faster than two LDS reads per column). */
    if (ac->fixed_length_ok && !ac->scan_dense && !(ac->scan_exact && !ac->scan_full_rows) && !flat_building && smh_alt_engine_depth == 0 &&
        ac->flat_parts != 1 && m * smh_keys_symbol_bits(alphabet) <= SMH_KEY_MAX_BITS) {
        const int *tsrc = ac->g_transition ? ac->g_transition : trans;
        const unsigned int *fsrc = ac->g_final ? ac->g_final : final;
        unsigned char *pats = ac_extract_patterns(tsrc, fsrc, ac->g_transition ? ac->states : R, alphabet, m, ac->finals);
        if (pats) ac->keys = smh_keys_build(pats, m, (int)ac->finals, alphabet, SMH_KEYS_LDS_BUDGET, NULL);
        free(pats);
    }
    return ac;

oom:
    smh_set_error("smh_ac_compile_tables: out of memory");
bad:
    free(order); free(depth); free(seen); free(leaf); free(canon); free(newid); free(full);
    if (ac && ref_mode == SMH_AC_REF_ADOPT) /* ownership passes only on success */
        ac->g_transition = NULL, ac->g_supply = NULL, ac->g_final = NULL;
    smh_ac_host_free(ac);
    return NULL;
}


/* ------------------------------------------------------------------ scan-table plan
 * The kernels never leave LDS in their inner loop, so the automaton the lanes walk must fit
 * there whole.  Cutting the DFA at depth K keeps exactly the rows [0, depth_first[K+1]) (BFS
 * numbering) and is again an Aho-Corasick automaton -- of the K-symbol prefixes.  Two layouts:
 * stride 1 (one lookup per symbol) and, for the 4-letter alphabet, stride 2 (one lookup per two
 * symbols).  Measured on MI355X (profiles/): stride 2 with K = m runs ~5.0 TB/s,
 * stride 1 ~3.6 TB/s (LDS lookup rate), stride 1 with K < m ~3.35 TB/s while candidates are rare;
 * stride 2 with K < m has to verify every candidate from the root and only pays when candidates are
 * very rare (it measured 2.0 TB/s at one candidate per 260 bytes).  Relative cost per text byte:
 *     stride 1:  1 + 0.08 [K < m] + 3 * P(a 16-byte piece of a wave holds a candidate)
 *     stride 2:  0.62 + 300 * r [K < m]                       r = candidates per text byte
 *     hybrid  :  0.51 (three chains per lane: halo <= 16 bytes; else 0.67) + 2.0 * P(a wave holds a lane deeper
 *                than D) + (0.07 + 300 * r) [K < m]
 *                (round 1 fit: K16D8 1.25, K16D7 2.26, K12D9 exact 0.91 / cut 0.98 of the stride-1 time; round 2
 *                refit of the constant terms after the leaner step and the dynamic chunk scheduling) */

static uint32_t entry_get(const void *t, int eb, size_t i)
{
    return eb == 2 ? ((const uint16_t *)t)[i] : ((const uint32_t *)t)[i];
}

/* target row and candidate flag of the depth-K automaton for (row, c) */
static inline uint32_t trunc_step(const struct smh_ac *ac, int K, uint32_t row, int c, int *flag)
{
    const uint32_t mask = ac->entry_bytes == 2 ? 0x7FFFu : 0x7FFFFFFFu;
    uint32_t e = entry_get(ac->table, ac->entry_bytes, (size_t)row * ac->alphabet + c);
    uint32_t t = e & mask;
    if (K >= ac->m) {
        *flag = (int)(e >> (ac->entry_bytes == 2 ? 15 : 31));
        return t;
    }
    if (ac->row_depth[t] > K) t = ac->row_fail[t]; /* depth K+1 -> its supply state, depth <= K */
    *flag = ac->row_depth[t] == K;
    return t;
}

static double candidate_rate(const struct smh_ac *ac, int K)
{
    if (K >= ac->m) return 0.0;
    double states_at_k = (double)(ac->depth_first[K + 1] - ac->depth_first[K]);
    double r = states_at_k;
    for (int i = 0; i < K; ++i) r /= (double)ac->alphabet;
    return r > 1.0 ? 1.0 : r;
}


/* ------------------------------------------------------------------ hybrid stride-2 table (alphabet 4)
 * A full stride-2 row costs 32 bytes, so only ~5000 rows fit LDS, while 1000 patterns of length 16
 * make ~11000.  But on random-looking text a lane is almost always in a SHALLOW state (the chance
 * of being deeper than d is (rows at depth d) / 4^d), and a deep state of a set of distinct
 * patterns is almost always a link of a chain: one child, one grandchild.  So:
 *   rows of depth <= D ("full")   16 entries of 16 bits: the row after the two symbols (no flags:
 *                                 D <= K-3, so two symbols from a full row never complete a prefix)
 *   deeper rows ("compact")       a list of 4-byte items, each "if the pair (or its first symbol)
 *                                 equals X then {flag F1/F2 | go to row N}", ending in the row's
 *                                 supply (fail) link.  Exact because for a pair x that is not a
 *                                 goto path out of s:  delta2(s, x) = delta2(fail(s), x), and when
 *                                 a goto path ends in an accepting (depth-K) leaf the automaton
 *                                 continues from that leaf's supply state, which again is
 *                                 delta2(fail(s), x)  (ac/ac.c:209-211 applied twice).
 * Item (32 bits): [0..15] link = where to continue when the item does not end the step (next item of
 * the same row, finally the supply row), [16..19] pair code c1*4+c2, [20..23] compare mask (15, or 12
 * to compare c1 only), [24..25] flags to raise on a hit (1: prefix ends at the first symbol, 2: at
 * the second), [26] hit => the step ends in row (this id + 1), [27] hit => the step ends in the row
 * whose id is in the next 4-byte slot.  Compact rows are numbered along their chains so that [26]
 * covers all but the branching states.  Row ids are 16 bits; full rows are 0 .. scan_full_rows - 1, the item slots of the
 * compact part SMH_HYB_COMPACT0 (0x8000) upwards. */
struct hyb_item { uint8_t code, c1only, flags; uint32_t target; /* old row id, or UINT32_MAX */ };

static uint32_t hyb_step(const struct smh_ac *ac, int K, uint32_t r, int c, int *flag)
{
    uint32_t t = trunc_step(ac, K, r, c, flag);
    if (K < ac->m && *flag) t = ac->row_fail[t]; /* depth-K rows are leaves of the cut automaton */
    return t;
}

static int hyb_items(const struct smh_ac *ac, int K, uint32_t s, struct hyb_item *it)
{
    int n = 0;
    const int d = ac->row_depth[s];
    for (int c1 = 0; c1 < 4; ++c1) {
        int f1;
        const uint32_t t1 = hyb_step(ac, K, s, c1, &f1);
        if (f1) {
            it[n++] = (struct hyb_item){(uint8_t)(c1 * 4), 1, 1, UINT32_MAX};
            continue;
        }
        if (ac->row_depth[t1] != d + 1) continue; /* not a goto edge */
        for (int c2 = 0; c2 < 4; ++c2) {
            int f2;
            const uint32_t t2 = hyb_step(ac, K, t1, c2, &f2);
            if (f2)
                it[n++] = (struct hyb_item){(uint8_t)(c1 * 4 + c2), 0, 2, UINT32_MAX};
            else if (ac->row_depth[t2] == d + 2)
                it[n++] = (struct hyb_item){(uint8_t)(c1 * 4 + c2), 0, 0, t2};
        }
    }
    /* a chain item first: it is the one that can use the implicit "next id" link */
    for (int i = 1; i < n; ++i)
        if (it[i].target != UINT32_MAX && it[0].target == UINT32_MAX) {
            struct hyb_item t = it[0]; it[0] = it[i]; it[i] = t;
            break;
        }
    return n;
}

/* returns SMH_OK and the image, SMH_EUNSUP when (K, D) does not fit or is not representable, HYB_ESLOTS when the
 * compact part needs more than 0x7FFF item slots (a smaller D only adds compact rows: the caller stops trying) */
#define HYB_ESLOTS (-100)
static int hyb_build(const struct smh_ac *ac, int K, int D, uint32_t budget, void **image, uint32_t *bytes, uint32_t *nf_out)
{
    if (ac->alphabet != 4 || K > ac->m || D < 1 || D > K - 3) return SMH_EUNSUP;
    const uint32_t rk = ac->depth_first[K], nf = ac->depth_first[D + 1];
    if (nf >= rk || (uint64_t)nf * 32u > budget) return SMH_EUNSUP;
    const uint32_t nc = rk - nf;
    uint32_t *hid = (uint32_t *)malloc((size_t)nc * sizeof(uint32_t));   /* new id of compact row nf + i */
    uint32_t *extra = (uint32_t *)malloc((size_t)nc * sizeof(uint32_t)); /* first pseudo slot of the row */
    uint32_t *order = (uint32_t *)malloc((size_t)nc * sizeof(uint32_t));
    int rc = SMH_EUNSUP;
    void *img = NULL;
    if (!hid || !extra || !order) { rc = SMH_ENOMEM; goto out; }
    memset(hid, 0xFF, (size_t)nc * sizeof(uint32_t));
    struct hyb_item it[16];
    uint32_t cursor = nf, n_order = 0;
    for (uint32_t s0 = nf; s0 < rk; ++s0) {
        if (hid[s0 - nf] != UINT32_MAX) continue;
        /* number the chain that starts here, then the extra items of its rows */
        const uint32_t first = n_order;
        for (uint32_t cur = s0;;) {
            hid[cur - nf] = cursor++;
            order[n_order++] = cur;
            const int n = hyb_items(ac, K, cur, it);
            if (n == 0 || it[0].target == UINT32_MAX || hid[it[0].target - nf] != UINT32_MAX) break;
            cur = it[0].target;
        }
        for (uint32_t k = first; k < n_order; ++k) {
            const uint32_t cur = order[k];
            const int n = hyb_items(ac, K, cur, it);
            extra[cur - nf] = cursor;
            /* item 0 sits in the row's own slot when it is a flag item or the implicit chain link */
            const int own = n > 0 && (it[0].target == UINT32_MAX || hid[it[0].target - nf] == hid[cur - nf] + 1u);
            for (int i = own ? 1 : 0; i < n; ++i) cursor += it[i].target == UINT32_MAX ? 1u : 2u;
        }
        if (cursor - nf > 0x7FFFu) { rc = HYB_ESLOTS; goto out; } /* compact ids are SMH_HYB_COMPACT0 + slot, 16 bits */
    }
    {
        const uint64_t total = (uint64_t)nf * 32u + (uint64_t)(cursor - nf) * 4u;
        if (total > budget) goto out;
        *bytes = (uint32_t)((total + 15u) & ~(uint64_t)15u);
    }
    img = calloc(*bytes + 16u, 1);
    if (!img) { rc = SMH_ENOMEM; goto out; }
    /* ids as the image holds them: full rows keep theirs, item slot s of the compact part is SMH_HYB_COMPACT0 + s (so that
     * the scan's unclamped full-row lookup of a compact id falls outside LDS: ac_lane.h smh_fmt_s2h) */
#define HYB_OUT(id) ((id) < nf ? (id) : (id) - nf + SMH_HYB_COMPACT0)
#define HYB_ID(r) HYB_OUT((r) < nf ? (r) : hid[(r) - nf])
    uint16_t *t16 = (uint16_t *)img;
    for (uint32_t r = 0; r < nf; ++r)
        for (int c1 = 0; c1 < 4; ++c1)
            for (int c2 = 0; c2 < 4; ++c2) {
                int f1, f2;
                const uint32_t r1 = hyb_step(ac, K, r, c1, &f1);
                const uint32_t r2 = hyb_step(ac, K, r1, c2, &f2);
                if (f1 || f2) goto out; /* cannot happen with D <= K-3 */
                t16[(size_t)r * 16 + c1 * 4 + c2] = (uint16_t)HYB_ID(r2);
            }
    uint32_t *rec = (uint32_t *)((uint8_t *)img + (size_t)nf * 32u); /* slot of id i: rec[i - nf] */
    for (uint32_t s = nf; s < rk; ++s) {
        const int n = hyb_items(ac, K, s, it);
        const uint32_t id = hid[s - nf], fail = HYB_ID(ac->row_fail[s]);
        const int own = n > 0 && (it[0].target == UINT32_MAX || hid[it[0].target - nf] == id + 1u);
        uint32_t item_slot[16], pos = extra[s - nf];
        for (int i = 0; i < n; ++i) {
            if (i == 0 && own) { item_slot[0] = id; continue; }
            item_slot[i] = pos;
            pos += it[i].target == UINT32_MAX ? 1u : 2u;
        }
        /* own slot without an item: a pass-through (mask 0 always "hits", but with no flags and no
         * end-of-step bits a hit just follows the link) */
        if (!own) rec[id - nf] = n > 0 ? HYB_OUT(item_slot[0]) : fail;
        for (int i = 0; i < n; ++i) {
            const uint32_t link = i == n - 1 ? fail : HYB_OUT(item_slot[i + 1]);
            uint32_t v = link | ((uint32_t)it[i].code << 16) | ((it[i].c1only ? 12u : 15u) << 20) |
                         ((uint32_t)it[i].flags << 24);
            if (it[i].target != UINT32_MAX) {
                if (i == 0 && own) {
                    v |= 1u << 26;
                } else {
                    v |= 1u << 27;
                    rec[item_slot[i] + 1u - nf] = HYB_OUT(hid[it[i].target - nf]);
                }
            }
            rec[item_slot[i] - nf] = v;
        }
    }
#undef HYB_ID
#undef HYB_OUT
    *image = img;
    img = NULL;
    *nf_out = nf;
    rc = SMH_OK;
out:
    free(hid); free(extra); free(order); free(img);
    return rc;
}

/* fraction of uniform-text positions at which the automaton is deeper than D */
static double deep_rate(const struct smh_ac *ac, int D)
{
    if (D + 1 > ac->max_depth) return 0.0;
    double r = (double)(ac->depth_first[D + 2] - ac->depth_first[D + 1]);
    for (int i = 0; i <= D; ++i) r /= 4.0;
    return r > 1.0 ? 1.0 : r;
}

/* dense plan tables: walk the full DFA over every m-symbol string (4^m <= 65536 of them) and note the accepting ones */
/* 0.164-0.166 ms/GiB measured for the pair lookup at one workgroup per CU (profiles/r03_q: configs[2] and the 8000 x 8 set)
 * against 0.289 for the exact stride-1 scan -- and 0.178-0.187 for the exact stride-2 image (cost 0.62), which it therefore
 * replaces as well: every 4-letter set of 3..8 symbols scans by the dense plan (until the occupancy fix it ran 0.20: 0.70) */
#define SMH_AC_DENSE_COST 0.57
static int dense_build(struct smh_ac *ac)
{
    const int m = ac->m;
    if (ac->alphabet != 4 || m < 3 || m > 8 || !ac->fixed_length_ok) return SMH_EUNSUP;
    if (ac->dense_pair) return SMH_OK;
    const uint32_t n_codes = 1u << (2 * m), mask = ac->entry_bytes == 2 ? 0x7FFFu : 0x7FFFFFFFu;
    uint32_t *filter = (uint32_t *)calloc(n_codes / 32u ? n_codes / 32u : 1u, sizeof(uint32_t));
    uint32_t *pair = (uint32_t *)calloc(16384u, sizeof(uint32_t));
    if (!filter || !pair) { free(filter); free(pair); return SMH_ENOMEM; }
    for (uint32_t code = 0; code < n_codes; ++code) { /* oldest symbol in the highest bits */
        uint32_t row = 0, e = 0;
        for (int i = m - 1; i >= 0; --i) {
            e = entry_get(ac->table, ac->entry_bytes, (size_t)row * 4u + ((code >> (2 * i)) & 3u));
            row = e & mask;
        }
        if (e >> (ac->entry_bytes == 2 ? 15 : 31)) filter[code >> 5] |= 1u << (code & 31u);
    }
    /* the same bits indexed by the code i of NINE symbols: the seven oldest select the dword, the newest pair two adjacent
     * bits in it -- "the m symbols ending at the 8th / at the 9th symbol are accepted" (smh_internal.h smh_wm.pair_table) */
    for (uint32_t i = 0; i < (1u << 18); ++i) {
        const uint32_t c1 = (i >> 2) & (n_codes - 1u), c2 = i & (n_codes - 1u), pr = i & 15u;
        if ((filter[c1 >> 5] >> (c1 & 31u)) & 1u) pair[i >> 4] |= 1u << (2u * pr);
        if ((filter[c2 >> 5] >> (c2 & 31u)) & 1u) pair[i >> 4] |= 1u << (2u * pr + 1u);
    }
    ac->dense_filter = filter;
    ac->dense_pair = pair;
    return SMH_OK;
}

static int plan_scan(struct smh_ac *ac, uint32_t lds_budget, int force_stride, int force_depth, int allow_hybrid)
{
    const int A = ac->alphabet;
    const int kmax = ac->m < SMH_AC_MAX_SCAN_DEPTH ? ac->m : SMH_AC_MAX_SCAN_DEPTH;
    const int no_dense = force_depth == -1; /* internal: the ordinary plan beside a dense one */
    if (no_dense) force_depth = 0;
    const int force_k = force_depth & 0xFF, force_d = (force_depth >> 8) & 0xFF;
    int best_k[4] = {0, 0, 0, 0}, best_d = 0;
    double best_cost = 1e30;
    int best_s = 0;
    for (int s = 1; s <= 2; ++s) {
        if (s == 2 && A != 4) continue;
        if (force_stride && force_stride != 4 && s != force_stride) continue;
        for (int K = kmax; K >= 1; --K) {
            if (force_k && K != force_k) continue;
            uint64_t rk = ac->depth_first[K + 1 <= ac->max_depth + 1 ? K + 1 : ac->max_depth + 1];
            if (K >= ac->m) rk = ac->rows;
            uint64_t per_row = s == 1 ? (uint64_t)A * (rk <= 32768 ? 2u : 4u) : (uint64_t)A * A * 2u;
            if (s == 2 && rk > 16384) continue;
            if (rk * per_row > lds_budget) continue;
            const double r = candidate_rate(ac, K);
            double cost;
            if (s == 1)
                cost = 1.0 + (K < ac->m ? 0.08 : 0.0) + (K > 17 ? 0.05 : 0.0) /* a halo beyond 16 bytes: measured 0.317 vs 0.300 ms/GiB */
                       + 3.0 * (1.0 - exp(-16.0 * 64.0 * r));
            else
                cost = 0.62 + 300.0 * r; /* round 2: 0.180 ms/GiB against 0.289 for the exact stride-1 scan */
            if (cost < best_cost) { best_cost = cost; best_s = s; best_k[s] = K; }
            if (!force_k) break; /* the deepest K that fits is the best for this stride */
        }
    }
    /* hybrid stride 2 (see hyb_build): for every K the deepest D whose estimated image fits.  Measured
     * (1000 patterns, m = 16 / 32, 1 GiB): the common step costs what a plain stride-2 step costs
     * (0.21 ms/GiB), a step in which ANY lane of the wave sits in a compact row 3.2-3.5 x that (two more
     * dependent LDS round trips and the item arithmetic), recording candidates as bits 0.04 ms/GiB */
    if (allow_hybrid && A == 4 && (!force_stride || force_stride == 3 || force_stride == 4)) {
        for (int K = kmax; K >= 4; --K) {
            if (force_k && K != force_k) continue;
            if (K < ac->m && K - 1 > 32) continue; /* the candidate-bit recording covers a 32-byte halo */
            for (int D = K - 3; D >= 1; --D) {
                if (force_d && D != force_d) continue;
                const uint64_t nf = ac->depth_first[D + 1], rk = ac->depth_first[K];
                if (nf >= rk) continue;
                const uint64_t nc = rk - nf;
                /* nc + nc/8 estimates the item slots (one per chain state, two more per branch): ids are 15 bits */
                if (nf * 32u + nc * 4u + nc / 4u + 64u > lds_budget || nc + nc / 8u > 0x7FFFu) continue;
                const double r = candidate_rate(ac, K), q = deep_rate(ac, D);
                /* round 2 refit (leaner step, dynamic chunk scheduling): K12D9 cut 0.205 ms/GiB with three chains per
                 * lane (halo <= 16 bytes; 0.220 with two), 0.255 with one, against 0.289 for the exact stride-1 scan */
                const double base = K - 1 <= 16 ? 0.51 : 0.67;
                /* round 3: + 7 q -- the resolution loop runs as long as ANY lane walks items, and with most lanes deep
                 * (3000 patterns of 16 symbols, K = 15 with rows to depth 5 only: q = 0.73) the image measured 2.18 ms/GiB
                 * = 7.5 units where "any lane deep" alone said 2.5; the stride-1 K = 11 plan of the same set: 0.62 */
                const double cost = base + 2.0 * (1.0 - pow(1.0 - q, 64.0)) + 7.0 * q + (K < ac->m ? 0.07 + 300.0 * r : 0.0);
                if (cost < best_cost) { best_cost = cost; best_s = 3; best_k[3] = K; best_d = D; }
                break;
            }
        }
    }
    /* dense plan: the complete 4^m-state automaton as a bit set (see smh_internal.h); stride code 4 forces it */
    int dense = 0;
    if (!no_dense && A == 4 && ac->m >= 3 && ac->m <= 8 && (force_stride == 4 || (!force_stride && !force_k && SMH_AC_DENSE_COST < best_cost))) {
        const int rc = dense_build(ac);
        if (rc == SMH_ENOMEM) { smh_set_error("smh_ac_plan_scan: out of memory"); return rc; }
        if (rc == SMH_OK) dense = 1;
    }
    if (force_stride == 4 && !dense) {
        smh_set_error("smh_ac_plan_scan: the dense plan needs alphabet 4 and 3 <= m <= 8");
        return SMH_EUNSUP;
    }
    if (dense && force_stride == 4) { /* the tables of an ordinary plan are still built: slow paths and positions fall back on them */
        const int rc = plan_scan(ac, lds_budget, 0, -1, allow_hybrid);
        if (rc != SMH_OK) return rc;
        ac->scan_dense = 1;
        ac->scan_cost = SMH_AC_DENSE_COST;
        return SMH_OK;
    }
    if (!best_s) {
        smh_set_error("smh_ac_plan_scan: no depth-K automaton fits %u bytes of LDS (alphabet %d)", lds_budget, A);
        return SMH_EUNSUP;
    }
    void *hyb_image = NULL;
    uint32_t hyb_bytes = 0, hyb_nf = 0;
    if (best_s == 3) {
        int rc = SMH_EUNSUP;
        for (int D = best_d; D >= 1 && rc != SMH_OK; --D) {
            rc = hyb_build(ac, best_k[3], D, lds_budget, &hyb_image, &hyb_bytes, &hyb_nf);
            if (rc == SMH_ENOMEM) { smh_set_error("smh_ac_plan_scan: out of memory"); return rc; }
            if (force_d || rc == HYB_ESLOTS) break; /* too many item slots: a smaller D has more of them */
        }
        if (rc != SMH_OK) {
            if (force_stride == 3) {
                smh_set_error("smh_ac_plan_scan: the hybrid stride-2 table does not fit %u bytes of LDS", lds_budget);
                return SMH_EUNSUP;
            }
            return plan_scan(ac, lds_budget, force_stride, force_depth, 0);
        }
    }
    const int K = best_k[best_s];
    const uint32_t rk = K >= ac->m ? ac->rows : ac->depth_first[K + 1];
    /* the new tables are built first and installed together: a failure leaves the handle's current plan intact */
    /* stride-1 depth-K table */
    const int eb1 = rk <= 32768 ? 2 : 4;
    const size_t n1 = (size_t)rk * A;
    void *t1 = calloc(n1 * eb1 + 16, 1);
    if (!t1) { free(hyb_image); smh_set_error("smh_ac_plan_scan: out of memory"); return SMH_ENOMEM; }
    for (uint32_t r = 0; r < rk; ++r)
        for (int c = 0; c < A; ++c) {
            int flag;
            uint32_t t = trunc_step(ac, K, r, c, &flag);
            if (eb1 == 2) ((uint16_t *)t1)[(size_t)r * A + c] = (uint16_t)(t | (flag ? 0x8000u : 0u));
            else ((uint32_t *)t1)[(size_t)r * A + c] = t | (flag ? 0x80000000u : 0u);
        }
    uint16_t *t2 = NULL;
    if (best_s == 2) {
        const size_t n2 = (size_t)rk * 16;
        t2 = (uint16_t *)calloc(n2 * 2 + 16, 1);
        if (!t2) { free(t1); smh_set_error("smh_ac_plan_scan: out of memory"); return SMH_ENOMEM; }
        for (uint32_t r = 0; r < rk; ++r)
            for (int c1 = 0; c1 < 4; ++c1)
                for (int c2 = 0; c2 < 4; ++c2) {
                    int f1, f2;
                    uint32_t r1 = trunc_step(ac, K, r, c1, &f1);
                    uint32_t r2 = trunc_step(ac, K, r1, c2, &f2);
                    t2[(size_t)r * 16 + c1 * 4 + c2] = (uint16_t)(r2 | (f1 ? 0x4000u : 0u) | (f2 ? 0x8000u : 0u));
                }
    }
    if (ac->trunc1_table != ac->scan_table) free(ac->trunc1_table);
    free(ac->scan_table);
    ac->scan_depth = K;
    ac->scan_stride = best_s == 3 ? 2 : best_s;
    ac->scan_full_rows = hyb_nf;
    ac->scan_exact = K >= ac->m;
    ac->scan_rows = rk;
    ac->scan_candidate_rate = candidate_rate(ac, K);
    ac->scan_cost = dense ? SMH_AC_DENSE_COST : best_cost;
    ac->scan_dense = dense;
    ac->trunc1_table = t1;
    ac->trunc1_entry_bytes = eb1;
    ac->trunc1_bytes = (uint64_t)n1 * eb1;
    if (best_s == 1) {
        ac->scan_table = t1;
        ac->scan_entry_bytes = eb1;
        ac->scan_bytes = (uint32_t)((n1 * eb1 + 15) & ~(size_t)15);
    } else if (best_s == 3) {
        ac->scan_table = hyb_image;
        ac->scan_entry_bytes = 2;
        ac->scan_bytes = hyb_bytes;
    } else {
        ac->scan_table = t2;
        ac->scan_entry_bytes = 2;
        ac->scan_bytes = (uint32_t)(((size_t)rk * 16 * 2 + 15) & ~(size_t)15);
    }
    return SMH_OK;
}

int smh_ac_plan_scan(struct smh_ac *ac, uint32_t lds_budget, int force_stride, int force_depth)
{
    return plan_scan(ac, lds_budget, force_stride, force_depth, 1);
}

smh_ac *smh_ac_compile_tables(const int *state_transition, const unsigned int *state_supply,
                              const unsigned int *state_final, uint64_t rows, int alphabet, int m)
{
    return smh_ac_compile_tables_impl(state_transition, state_supply, state_final, rows, alphabet, m,
                                      SMH_AC_REF_COPY);
}

smh_ac *smh_ac_compile_patterns(const unsigned char *pattern_flat, int m, int p_size, int alphabet)
{
    if (!pattern_flat || m < 1 || p_size < 1 || alphabet < 1 || alphabet > 256) {
        smh_set_error("smh_ac_compile_patterns: bad arguments");
        return NULL;
    }
    for (size_t i = 0; i < (size_t)m * p_size; ++i)
        if ((int)pattern_flat[i] >= alphabet) {
            smh_set_error("smh_ac_compile_patterns: symbol %u >= alphabet %d", pattern_flat[i], alphabet);
            return NULL;
        }
    /* the reference-layout tables, sized and initialised as main.c:410-420 does */
    size_t rows = (size_t)m * p_size + 1;
    int *trans = (int *)malloc(rows * alphabet * sizeof(int));
    unsigned int *supply = (unsigned int *)calloc(rows, sizeof(unsigned int));
    unsigned int *final = (unsigned int *)calloc(rows, sizeof(unsigned int));
    unsigned char **ptrs = (unsigned char **)malloc((size_t)p_size * sizeof(unsigned char *));
    if (!trans || !supply || !final || !ptrs) {
        free(trans); free(supply); free(final); free(ptrs);
        smh_set_error("smh_ac_compile_patterns: out of memory");
        return NULL;
    }
    memset(trans, -1, rows * alphabet * sizeof(int));
    for (int j = 0; j < p_size; ++j) ptrs[j] = (unsigned char *)pattern_flat + (size_t)j * m;
    uint32_t idcounter, patterncounter;
    ac_fill_tables(ptrs, m, p_size, alphabet, trans, supply, final, &idcounter, &patterncounter);
    free(ptrs);
    smh_ac *ac = smh_ac_compile_tables_impl(trans, supply, final, (uint64_t)idcounter, alphabet, m,
                                            SMH_AC_REF_ADOPT);
    if (!ac) { free(trans); free(supply); free(final); }
    return ac;
}

int smh_ac_get_info(const smh_ac *ac, smh_ac_info *out)
{
    if (!ac || ac->magic != SMH_MAGIC_AC || !out) {
        smh_set_error("smh_ac_get_info: bad handle");
        return SMH_EINVAL;
    }
    memset(out, 0, sizeof *out);
    out->alphabet = (uint32_t)ac->alphabet;
    out->m = (uint32_t)ac->m;
    out->states = ac->states;
    out->finals = ac->finals;
    out->rows = ac->rows;
    out->entry_bytes = (uint32_t)ac->entry_bytes;
    out->table_bytes = ac->table_bytes;
    out->lds_rows = ac->scan_rows;
    out->lds_bytes = ac->scan_bytes;
    out->scan_depth = (uint32_t)ac->scan_depth;
    out->scan_stride = (uint32_t)ac->scan_stride;
    out->scan_exact = (uint32_t)ac->scan_exact;
    out->scan_full_rows = ac->scan_full_rows;
    out->scan_engine = ac->engine_forced >= 0 ? (uint32_t)ac->engine_forced : (ac->alt_wm ? SMH_ALGO_WM : SMH_ALGO_AC);
    out->scan_dense = (uint32_t)ac->scan_dense;
    out->flat_parts = (uint32_t)ac->flat_parts;
    out->key_slots = ac->keys ? 2u * ac->keys->P.slots : 0u;
    out->hash_slots = smh_ac_hash_engine(ac) ? 4u * smh_ac_hash_engine(ac)->P.slots : 0u;
    out->adaptive = (ac->flex_wm || ac->flat_ac || ac->keys || out->hash_slots) && ac->engine_forced < 0 ? 1u : 0u;
    if (out->scan_engine == SMH_ALGO_WM) {
        smh_wm_info wi;
        if (smh_wm_get_info(ac->alt_wm ? ac->alt_wm : ac->flex_wm, &wi) == SMH_OK) {
            out->verify_in_registers = wi.verify_in_registers;
            out->gram_kind = wi.gram_kind;
        }
    }
    return SMH_OK;
}

int smh_ac_set_scan_plan(smh_ac *ac, int stride, int depth)
{
    if (!ac || ac->magic != SMH_MAGIC_AC || stride < 0 || stride > 4 || depth < 0) {
        smh_set_error("smh_ac_set_scan_plan: bad arguments");
        return SMH_EINVAL;
    }
    if (!ac->fixed_length_ok) {
        smh_set_error("smh_ac_set_scan_plan: patterns are not all of length m");
        return SMH_EUNSUP;
    }
    const int rc = smh_ac_plan_scan(ac, SMH_AC_LDS_BUDGET, stride, depth);
    if (rc != SMH_OK) return rc; /* the handle keeps its current plan and device tables */
    if (ac->dev) smh_ac_dev_free(ac->dev); /* device copies are rebuilt on the next scan */
    ac->dev = NULL;
    smh_adapt_dev_free(ac->adapt); /* what was measured belongs to the old plan */
    ac->adapt = NULL;
    ac->alt_off = stride != 0 || depth != 0; /* a forced plan means "run the automaton kernels" */
    ac->engine_forced = ac->alt_off ? SMH_ALGO_AC : -1;
    if (ac->scan_exact || ac->scan_dense) ac->flex_wm = NULL; /* an exact plan never hands over */
    else if (ac->hv_wm && !ac->hv_wm->alt_ac && !ac->hv_wm->pair_table && (ac->hv_wm->gram_kind != SMH_GRAM_NONE || !ac->hv_wm->filter_exact) &&
             ac->hv_wm->scan_ms_est > 0)
        ac->flex_wm = ac->hv_wm;
    ++ac->generation;
    return SMH_OK;
}

/* the window-hash engine an automaton handle runs as SMH_ENGINE_HASH: its filter engine's, unless it has a key table of its own */
struct smh_hashes *smh_ac_hash_engine(const struct smh_ac *ac)
{
    const struct smh_wm *fw = ac->alt_wm ? ac->alt_wm : ac->flex_wm;
    return fw && !ac->keys ? fw->hashes : NULL;
}

int smh_ac_set_scan_engine(smh_ac *ac, int engine)
{
    if (!ac || ac->magic != SMH_MAGIC_AC || (engine != -1 && engine != SMH_ALGO_WM && engine != SMH_ALGO_AC && engine != SMH_ENGINE_AC_FLAT && engine != SMH_ENGINE_KEYS && engine != SMH_ENGINE_HASH)) {
        smh_set_error("smh_ac_set_scan_engine: bad arguments");
        return SMH_EINVAL;
    }
    if (engine == SMH_ENGINE_HASH && !smh_ac_hash_engine(ac)) {
        smh_set_error("smh_ac_set_scan_engine: this handle keeps no window-hash engine (alphabet 4, a key table, an exact plan, or m outside 4..32)");
        return SMH_EUNSUP;
    }
    if (engine == SMH_ENGINE_KEYS && !ac->keys) {
        smh_set_error("smh_ac_set_scan_engine: this handle keeps no key table (m * bits per symbol > 64, more keys than LDS holds, or its plan is an exact one-launch plan)");
        return SMH_EUNSUP;
    }
    if (engine == SMH_ENGINE_AC_FLAT && !ac->flat_ac) {
        smh_set_error("smh_ac_set_scan_engine: this set keeps no plain stride-1 automata (its plan is exact and plain already, or more than %d parts would be needed)", SMH_FLAT_MAX_PARTS);
        return SMH_EUNSUP;
    }
    if (engine == SMH_ALGO_WM && !ac->alt_wm && !ac->flex_wm) {
        smh_set_error("smh_ac_set_scan_engine: this set has no suffix-filter engine (its automaton plan is exact)");
        return SMH_EUNSUP;
    }
    ac->alt_off = engine == SMH_ALGO_AC;
    ac->engine_forced = engine;
    ++ac->generation;
    return SMH_OK;
}

struct smh_ac *smh_ac_flat_part(struct smh_ac *ac, int i)
{
    if (!ac || ac->magic != SMH_MAGIC_AC || i < 0) return NULL;
    struct smh_ac *p = ac->flat_ac;
    while (p && i-- > 0) p = p->flat_next;
    return p;
}

void smh_ac_free(smh_ac *ac)
{
    if (!ac) return;
    if (ac->dev) smh_ac_dev_free(ac->dev);
    ac->dev = NULL;
    smh_adapt_dev_free(ac->adapt);
    ac->adapt = NULL;
    smh_ac_host_free(ac);
}
