/*
 * csrc/sh_host.c -- Set-Horspool, host side (first sibling algorithm).
 *
 *   preBmBc      the set-Horspool bad-character table.  The reference calls it (main.c:173) but its
 *                definition is in the missing ../helper.o; this is the textbook table:
 *                bmBc[c] = min over patterns, over i < m-1 with p[i] == c, of m-1-i, else m.
 *   preproc_sh   sh/sh.c:178-196 -- the patterns inserted BACKWARDS into a trie, written to the
 *                caller's flat state_transition / state_final exactly as the reference numbers it
 *   search_sh    sh/sh.c:151-176 -- same count on the GPU
 *   free_sh      sh/sh.c:198-203
 *   smh_sh_*     handle-based superset (include/smatcher_hip.h)
 *
 * What search_sh computes: at every column the skip loop visits, the reversed trie is walked from
 * text[column] backwards; the column counts when the walk ends in a final state.  A VALID
 * bad-character table only skips columns at which no pattern can end, so the count is the number of
 * end columns of pattern occurrences -- the quantity search_ac and search_wu return.  Two device
 * paths (smh_runtime.hip):
 *   SMH_VARIANT_TABLE  sh_table_kernel: the reference-layout reversed trie walked as given, with
 *                      the caller's bmBc driving a per-lane skip loop (cuda/cuda_sh.cu:23-108)
 *   SMH_VARIANT_TUNED  walking a reversed trie from the end of the window IS "test the window's
 *                      last W symbols, then the rest": the first W levels of the reversed trie are
 *                      the Wu-Manber block filter (a block code is set iff that W-deep path exists)
 *                      and the remaining levels are its verify stage.  The patterns are read back
 *                      from the trie and scanned by the same tuned kernels (wm_block_kernel /
 *                      wm_pair_kernel; ac_dfa_kernel when Wu-Manber cannot take the set: m < 3 or an
 *                      alphabet outside wu/wu.c:18-47).
 * A bmBc entry larger than the valid table's would make the reference skip real matches depending
 * on where its loop happens to start; such a table is refused rather than imitated.
 */
#include "smh_internal.h"

#include <stdlib.h>
#include <string.h>

void preBmBc(unsigned char **pattern, int m, int p_size, int alphabet, int *bmBc)
{
    if (!pattern || !bmBc || m < 1 || p_size < 0 || alphabet < 1 || alphabet > 256) fail("preBmBc: bad arguments\n");
    for (int c = 0; c < alphabet; ++c) bmBc[c] = m;
    for (int i = 0; i < p_size; ++i)
        for (int j = 0; j + 1 < m; ++j) {
            const unsigned c = pattern[i][j];
            if ((int)c >= alphabet) fail("preBmBc: pattern symbol outside the alphabet\n");
            if (m - 1 - j < bmBc[c]) bmBc[c] = m - 1 - j;
        }
}

/* the reversed trie in the caller's flat table; returns the number of states */
static uint32_t sh_fill_tables(unsigned char **pattern, int m, int p_size, int alphabet, int *state_transition,
                               unsigned int *state_final, uint32_t *patterncounter_out)
{
    const size_t A = (size_t)alphabet;
    /* sh_init, sh/sh.c:37-63: row 0 all zero */
    for (int c = 0; c < alphabet; ++c) state_transition[c] = 0;
    uint32_t idcounter = 1, patterncounter = 0;
    /* sh_addstring, sh/sh.c:82-149: last symbol first; new state id = creation order */
    for (int i = 0; i < p_size; ++i) {
        const unsigned char *s = pattern[i];
        uint32_t state = 0;
        for (int j = m - 1; j >= 0; --j) {
            const unsigned c = s[j];
            if ((int)c >= alphabet) fail("preproc_sh: pattern symbol outside the alphabet\n");
            int32_t nx = state_transition[state * A + c];
            if (nx <= 0) { /* 0 in row 0, -1 elsewhere: no edge yet */
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
    *patterncounter_out = patterncounter;
    return idcounter;
}

void smh_sh_host_free(struct smh_sh *sh)
{
    if (!sh) return;
    free(sh->g_transition);
    free(sh->g_final);
    free(sh->patterns);
    free(sh->valid_bmbc);
    smh_wm_free(sh->wm);
    smh_ac_free(sh->ac);
    sh->magic = 0;
    free(sh);
}

/* validate the reversed trie, read the patterns back from it (depth-first, so each pattern is the
 * reversed path to a final depth-m state), build the valid bmBc and the tuned engine */
static struct smh_sh *sh_compile(const int *trans, const unsigned int *final, uint64_t rows_in, int alphabet, int m)
{
    if (!trans || !final || rows_in < 1 || alphabet < 1 || alphabet > 256 || m < 1 || rows_in > 0x7FFFFFFFull) {
        smh_set_error("smh_sh_compile_tables: bad arguments");
        return NULL;
    }
    const size_t A = (size_t)alphabet;
    const uint32_t R = (uint32_t)rows_in;
    struct smh_sh *sh = (struct smh_sh *)calloc(1, sizeof *sh);
    uint32_t *stack = (uint32_t *)malloc(((size_t)m + 2) * sizeof(uint32_t)); /* state per depth */
    int *sym = (int *)malloc(((size_t)m + 2) * sizeof(int));                   /* next symbol to try per depth */
    uint8_t *seen = (uint8_t *)calloc(R, 1);
    unsigned char *path = (unsigned char *)malloc((size_t)m + 1);
    if (!sh || !stack || !sym || !seen || !path) goto oom;
    sh->magic = SMH_MAGIC_SH;
    sh->alphabet = alphabet;
    sh->m = m;
    size_t cap = 64, np = 0;
    sh->patterns = (unsigned char *)malloc(cap * (size_t)m);
    if (!sh->patterns) goto oom;
    uint32_t max_id = 0, finals = 0;
    int depth = 0;
    stack[0] = 0;
    sym[0] = 0;
    seen[0] = 1;
    while (depth >= 0) {
        const uint32_t r = stack[depth];
        if (sym[depth] == 0 && final[r]) {
            if (depth != m) {
                smh_set_error("smh_sh_compile_tables: final state %u at depth %d (patterns must all have length m = %d)", r, depth, m);
                goto bad;
            }
            ++finals;
            if (np == cap) {
                cap *= 2;
                unsigned char *q = (unsigned char *)realloc(sh->patterns, cap * (size_t)m);
                if (!q) goto oom;
                sh->patterns = q;
            }
            for (int j = 0; j < m; ++j) sh->patterns[np * (size_t)m + (size_t)j] = path[m - 1 - j]; /* path is last-symbol-first */
            ++np;
        }
        int c = sym[depth];
        int32_t s = -1;
        for (; c < alphabet && depth < m; ++c) {
            s = trans[r * A + (size_t)c];
            if (s > 0) break;
        }
        if (depth >= m || c >= alphabet) {
            --depth;
            continue;
        }
        sym[depth] = c + 1;
        if ((uint32_t)s >= R || seen[s]) {
            smh_set_error("smh_sh_compile_tables: state_transition is not a trie (edge %u -> %d)", r, s);
            goto bad;
        }
        seen[s] = 1;
        if ((uint32_t)s > max_id) max_id = (uint32_t)s;
        path[depth] = (unsigned char)c;
        ++depth;
        stack[depth] = (uint32_t)s;
        sym[depth] = 0;
    }
    if (np == 0) {
        smh_set_error("smh_sh_compile_tables: the trie holds no pattern");
        goto bad;
    }
    sh->states = max_id + 1u;
    sh->finals = finals;
    sh->n_patterns = (uint32_t)np;
    /* reference-layout copy for the table-walking kernel, truncated to the ids in use */
    sh->g_transition = (int32_t *)malloc((size_t)sh->states * A * sizeof(int32_t));
    sh->g_final = (uint32_t *)malloc((size_t)sh->states * sizeof(uint32_t));
    sh->valid_bmbc = (int32_t *)malloc(A * sizeof(int32_t));
    if (!sh->g_transition || !sh->g_final || !sh->valid_bmbc) goto oom;
    memcpy(sh->g_transition, trans, (size_t)sh->states * A * sizeof(int32_t));
    for (uint32_t u = 0; u < sh->states; ++u) sh->g_final[u] = seen[u] && final[u] ? 1u : 0u;
    for (int c = 0; c < alphabet; ++c) sh->valid_bmbc[c] = m;
    for (size_t i = 0; i < np; ++i)
        for (int j = 0; j + 1 < m; ++j) {
            const unsigned c = sh->patterns[i * (size_t)m + (size_t)j];
            if (m - 1 - j < sh->valid_bmbc[c]) sh->valid_bmbc[c] = m - 1 - j;
        }
    /* tuned engine: the Wu-Manber kernels when they can take the set, else the automaton kernels */
    if (m >= 3 && smh_wu_shiftsize_for(alphabet))
        sh->wm = smh_wm_compile(sh->patterns, m, (int)np, alphabet);
    else
        sh->ac = smh_ac_compile_patterns(sh->patterns, m, (int)np, alphabet);
    if (!sh->wm && !sh->ac) goto bad; /* the engine's compiler has set the error text */
    free(stack); free(sym); free(seen); free(path);
    return sh;
oom:
    smh_set_error("smh_sh_compile_tables: out of memory");
bad:
    free(stack); free(sym); free(seen); free(path);
    if (sh) { sh->magic = SMH_MAGIC_SH; smh_sh_host_free(sh); }
    return NULL;
}

smh_sh *smh_sh_compile_tables(const int *state_transition, const unsigned int *state_final, uint64_t rows,
                              int alphabet, int m)
{
    return sh_compile(state_transition, state_final, rows, alphabet, m);
}

smh_sh *smh_sh_compile_patterns(const unsigned char *pattern_flat, int m, int p_size, int alphabet)
{
    if (!pattern_flat || m < 1 || p_size < 1 || alphabet < 1 || alphabet > 256) {
        smh_set_error("smh_sh_compile_patterns: bad arguments");
        return NULL;
    }
    for (size_t i = 0; i < (size_t)m * p_size; ++i)
        if ((int)pattern_flat[i] >= alphabet) {
            smh_set_error("smh_sh_compile_patterns: symbol %u >= alphabet %d", pattern_flat[i], alphabet);
            return NULL;
        }
    const size_t rows = (size_t)m * p_size + 1;
    int *trans = (int *)malloc(rows * alphabet * sizeof(int));
    unsigned int *final = (unsigned int *)calloc(rows, sizeof(unsigned int));
    unsigned char **ptrs = (unsigned char **)malloc((size_t)p_size * sizeof(unsigned char *));
    if (!trans || !final || !ptrs) {
        free(trans); free(final); free(ptrs);
        smh_set_error("smh_sh_compile_patterns: out of memory");
        return NULL;
    }
    memset(trans, -1, rows * alphabet * sizeof(int));
    for (int j = 0; j < p_size; ++j) ptrs[j] = (unsigned char *)pattern_flat + (size_t)j * m;
    uint32_t pc;
    const uint32_t idcounter = sh_fill_tables(ptrs, m, p_size, alphabet, trans, final, &pc);
    smh_sh *sh = sh_compile(trans, final, idcounter, alphabet, m);
    free(trans); free(final); free(ptrs);
    return sh;
}

int smh_sh_get_info(const smh_sh *sh, smh_sh_info *out)
{
    if (!sh || sh->magic != SMH_MAGIC_SH || !out) {
        smh_set_error("smh_sh_get_info: bad handle");
        return SMH_EINVAL;
    }
    memset(out, 0, sizeof *out);
    out->alphabet = (uint32_t)sh->alphabet;
    out->m = (uint32_t)sh->m;
    out->states = sh->states;
    out->finals = sh->finals;
    out->tuned_engine = sh->wm ? SMH_ALGO_WM : SMH_ALGO_AC;
    return SMH_OK;
}

int smh_sh_valid_bmbc(const smh_sh *sh, int *bmBc)
{
    if (!sh || sh->magic != SMH_MAGIC_SH || !bmBc) {
        smh_set_error("smh_sh_valid_bmbc: bad arguments");
        return SMH_EINVAL;
    }
    memcpy(bmBc, sh->valid_bmbc, (size_t)sh->alphabet * sizeof(int));
    return SMH_OK;
}

/* 1 <= bmBc[c] <= the valid shift for every symbol; NULL means "use the valid table" */
int smh_sh_check_bmbc(const struct smh_sh *sh, const int *bmBc)
{
    if (!bmBc) return SMH_OK;
    for (int c = 0; c < sh->alphabet; ++c)
        if (bmBc[c] < 1 || bmBc[c] > sh->valid_bmbc[c]) {
            smh_set_error("bmBc[%d] = %d is not a valid set-Horspool shift for these patterns (1..%d): the reference "
                          "would skip matches depending on where its loop starts", c, bmBc[c], sh->valid_bmbc[c]);
            return SMH_EINVAL;
        }
    return SMH_OK;
}

void smh_sh_free(smh_sh *sh)
{
    if (!sh || sh->magic != SMH_MAGIC_SH) return;
    if (sh->dev) smh_sh_dev_free(sh->dev);
    sh->dev = NULL;
    smh_sh_host_free(sh);
}

/* ------------------------------------------------------------------ legacy names */
struct ac_table *preproc_sh(unsigned char **pattern, int m, int p_size, int alphabet, int *state_transition,
                            unsigned int *state_final)
{
    if (m < 1 || p_size < 0 || alphabet < 1 || alphabet > 256) fail("preproc_sh: bad arguments\n");
    struct smh_sh_table_box *box = (struct smh_sh_table_box *)calloc(1, sizeof *box);
    if (!box) fail("Could not initialize table\n");
    uint32_t pc = 0;
    const uint32_t idcounter = sh_fill_tables(pattern, m, p_size, alphabet, state_transition, state_final, &pc);
    box->pub.idcounter = idcounter;
    box->pub.patterncounter = pc;
    box->pub.zerostate = NULL; /* the pointer trie of the reference is not materialised */
    box->magic = SMH_MAGIC_SH;
    box->sh = sh_compile(state_transition, state_final, idcounter, alphabet, m);
    if (!box->sh) {
        fputs(smh_last_error(), stderr);
        fail("\npreproc_sh: could not compile the reversed trie\n");
    }
    return &box->pub;
}

void free_sh(struct ac_table *table, int alphabet)
{
    (void)alphabet;
    if (!table) return;
    struct smh_sh_table_box *box = (struct smh_sh_table_box *)table;
    if (box->magic != SMH_MAGIC_SH) fail("free_sh: not a table from preproc_sh\n");
    smh_sh_free(box->sh);
    box->magic = 0;
    free(box);
}
