/*
 * csrc/sbom_host.c -- Set Backward Oracle Matching, host side (second sibling).
 *
 *   preproc_sbom  sbom/sbom.c:198-214 -- the factor oracle of the REVERSED patterns, built pattern by
 *                 pattern: trie edges plus the oracle's external transitions, written to the caller's
 *                 flat state_transition, and per state the list {count, pattern ids...} of the
 *                 patterns that end there, in 200-entry rows of state_final_multi (main.c:422-425)
 *   search_sbom   sbom/sbom.c:128-172 -- same count on the GPU
 *   free_sbom     sbom/sbom.c:216-236
 *   smh_sbom_*    handle-based superset (include/smatcher_hip.h)
 *
 * What search_sbom computes: at every column its loop visits, the oracle is walked from text[column]
 * backwards; when all m symbols are read and the state lists patterns, they are compared with the
 * window and the column counts once if one is equal; otherwise the loop skips max(m - j, 1) columns,
 * which the oracle guarantees end no occurrence.  So the count is again the number of end columns of
 * pattern occurrences.  Two device paths (smh_runtime.hip):
 *   SMH_VARIANT_TABLE  sbom_table_kernel: oracle and lists walked from HBM/L2 as given, the
 *                      reference's loop per lane (cuda/cuda_sbom.cu:23-123)
 *   SMH_VARIANT_TUNED  the oracle only FILTERS (it accepts more than the patterns' factors, hence the
 *                      reference's memcmp); filter-then-compare over a window read backwards is what
 *                      the Wu-Manber kernels do with a suffix block, so the patterns are handed to
 *                      that engine (the automaton kernels when Wu-Manber cannot take the set).
 */
#include "smh_internal.h"

#include <stdlib.h>
#include <string.h>

struct sbom_state **pointer_array = NULL; /* smatcher.h:55: allocated and freed by the caller (main.c:208,229); unused here */

#define SBOM_NONE 0xFFFFFFFFu

static inline int sbom_edge(const int *t, size_t A, uint32_t state, unsigned c) { return t[state * A + c] > 0; }

/* returns the number of states, 0 when a state_final_multi row would overflow (more than 199 patterns
 * ending in one state: the reference writes past its 200-entry row there); *patterncounter_out =
 * patterns appended (duplicates included) */
static uint32_t sbom_fill_tables(unsigned char *const *rows, const unsigned char *flat, int m, int p_size, int alphabet,
                                 int *state_transition, unsigned int *state_final_multi, uint32_t *patterncounter_out)
{
    const size_t A = (size_t)alphabet;
    uint32_t *supply = (uint32_t *)malloc(((size_t)m * p_size + 1) * sizeof(uint32_t));
    if (!supply) fail("Could not allocate memory\n");
    supply[0] = SBOM_NONE; /* the root has no supply state (sbom/sbom.c:36) */
    for (int c = 0; c < alphabet; ++c) state_transition[c] = 0;
    uint32_t idcounter = 1, patterncounter = 0;
    for (int i = 0; i < p_size; ++i) {
        const unsigned char *s = rows ? rows[i] : flat + (size_t)i * m;
        for (int j = 0; j < m; ++j)
            if ((int)s[j] >= alphabet) fail("preproc_sbom: pattern symbol outside the alphabet\n");
        uint32_t state = 0;
        int j = m - 1;
        /* follow whatever exists, external transitions included (sbom/sbom.c:62-72) */
        while (j >= 0 && sbom_edge(state_transition, A, state, s[j])) {
            state = (uint32_t)state_transition[state * A + s[j]];
            --j;
        }
        for (; j >= 0; --j) {
            const unsigned c = s[j];
            const uint32_t next = idcounter++;
            state_transition[state * A + c] = (int)next;
            uint32_t k = supply[state];
            while (k != SBOM_NONE && !sbom_edge(state_transition, A, k, c)) {
                state_transition[k * A + c] = (int)next; /* external transition */
                k = supply[k];
            }
            supply[next] = k != SBOM_NONE ? (uint32_t)state_transition[k * A + c] : 0u;
            state = next;
        }
        unsigned int *row = state_final_multi + (size_t)state * 200;
        if (row[0] >= 199u) {
            free(supply);
            *patterncounter_out = patterncounter;
            return 0;
        }
        row[row[0] + 1] = patterncounter++;
        row[0] += 1;
    }
    free(supply);
    *patterncounter_out = patterncounter;
    return idcounter;
}

void smh_sbom_host_free(struct smh_sbom *sb)
{
    if (!sb) return;
    free(sb->g_transition);
    free(sb->g_final_off);
    free(sb->g_final_ids);
    free(sb->patterns);
    smh_wm_free(sb->wm);
    smh_ac_free(sb->ac);
    sb->magic = 0;
    free(sb);
}

/* from the filled tables and the patterns: validated copies for the table-walking kernel (the
 * 200-entry rows packed into offsets + ids) and the tuned engine */
static struct smh_sbom *sbom_compile(const unsigned char *pattern_flat, int m, int p_size, int alphabet,
                                     const int *trans, const unsigned int *final_multi, uint64_t rows_in)
{
    if (!pattern_flat || !trans || !final_multi || m < 1 || p_size < 1 || alphabet < 1 || alphabet > 256 ||
        rows_in < 1 || rows_in > 0x7FFFFFFFull) {
        smh_set_error("smh_sbom_compile_tables: bad arguments");
        return NULL;
    }
    const size_t A = (size_t)alphabet;
    for (size_t i = 0; i < (size_t)m * p_size; ++i)
        if ((int)pattern_flat[i] >= alphabet) {
            smh_set_error("smh_sbom_compile: symbol %u >= alphabet %d", pattern_flat[i], alphabet);
            return NULL;
        }
    /* ids in use: every edge target must be a row of the table */
    uint32_t states = 1;
    for (uint32_t r = 0; r < states && r < rows_in; ++r)
        for (int c = 0; c < alphabet; ++c) {
            const int s = trans[r * A + (size_t)c];
            if (s > 0) {
                if ((uint64_t)s >= rows_in) {
                    smh_set_error("smh_sbom_compile_tables: edge %u -> %d leaves the table", r, s);
                    return NULL;
                }
                if ((uint32_t)s + 1u > states) states = (uint32_t)s + 1u;
            }
        }
    struct smh_sbom *sb = (struct smh_sbom *)calloc(1, sizeof *sb);
    if (!sb) goto oom;
    sb->magic = SMH_MAGIC_SBOM;
    sb->alphabet = alphabet;
    sb->m = m;
    sb->n_patterns = (uint32_t)p_size;
    sb->states = states;
    sb->g_transition = (int32_t *)malloc((size_t)states * A * sizeof(int32_t));
    sb->g_final_off = (uint32_t *)malloc(((size_t)states + 1) * sizeof(uint32_t));
    sb->patterns = (unsigned char *)malloc((size_t)m * p_size);
    if (!sb->g_transition || !sb->g_final_off || !sb->patterns) goto oom;
    memcpy(sb->g_transition, trans, (size_t)states * A * sizeof(int32_t));
    memcpy(sb->patterns, pattern_flat, (size_t)m * p_size);
    uint64_t total = 0;
    for (uint32_t r = 0; r < states; ++r) {
        const unsigned int cnt = final_multi[(size_t)r * 200];
        if (cnt > 199u) {
            smh_set_error("smh_sbom_compile_tables: state %u lists %u patterns (rows hold 199)", r, cnt);
            goto bad;
        }
        sb->g_final_off[r] = (uint32_t)total;
        total += cnt;
    }
    sb->g_final_off[states] = (uint32_t)total;
    sb->listed = (uint32_t)total;
    sb->g_final_ids = (uint32_t *)malloc((total ? total : 1) * sizeof(uint32_t));
    if (!sb->g_final_ids) goto oom;
    for (uint32_t r = 0; r < states; ++r)
        for (uint32_t i = 0; i < sb->g_final_off[r + 1] - sb->g_final_off[r]; ++i) {
            const unsigned int id = final_multi[(size_t)r * 200 + 1 + i];
            if (id >= (unsigned int)p_size) {
                smh_set_error("smh_sbom_compile_tables: state %u lists pattern %u of %d", r, id, p_size);
                goto bad;
            }
            sb->g_final_ids[sb->g_final_off[r] + i] = id;
        }
    if (m >= 3 && smh_wu_shiftsize_for(alphabet))
        sb->wm = smh_wm_compile(sb->patterns, m, p_size, alphabet);
    else
        sb->ac = smh_ac_compile_patterns(sb->patterns, m, p_size, alphabet);
    if (!sb->wm && !sb->ac) goto bad;
    return sb;
oom:
    smh_set_error("smh_sbom_compile: out of memory");
bad:
    if (sb) { sb->magic = SMH_MAGIC_SBOM; smh_sbom_host_free(sb); }
    return NULL;
}

smh_sbom *smh_sbom_compile_tables(const unsigned char *pattern_flat, int m, int p_size, int alphabet,
                                  const int *state_transition, const unsigned int *state_final_multi, uint64_t rows)
{
    return sbom_compile(pattern_flat, m, p_size, alphabet, state_transition, state_final_multi, rows);
}

smh_sbom *smh_sbom_compile_patterns(const unsigned char *pattern_flat, int m, int p_size, int alphabet)
{
    if (!pattern_flat || m < 1 || p_size < 1 || alphabet < 1 || alphabet > 256) {
        smh_set_error("smh_sbom_compile_patterns: bad arguments");
        return NULL;
    }
    for (size_t i = 0; i < (size_t)m * p_size; ++i)
        if ((int)pattern_flat[i] >= alphabet) {
            smh_set_error("smh_sbom_compile_patterns: symbol %u >= alphabet %d", pattern_flat[i], alphabet);
            return NULL;
        }
    const size_t rows = (size_t)m * p_size + 1;
    int *trans = (int *)malloc(rows * alphabet * sizeof(int));
    unsigned int *fm = (unsigned int *)calloc(rows * 200, sizeof(unsigned int));
    if (!trans || !fm) {
        free(trans); free(fm);
        smh_set_error("smh_sbom_compile_patterns: out of memory");
        return NULL;
    }
    memset(trans, -1, rows * alphabet * sizeof(int));
    uint32_t pc;
    const uint32_t idcounter = sbom_fill_tables(NULL, pattern_flat, m, p_size, alphabet, trans, fm, &pc);
    if (!idcounter) {
        free(trans); free(fm);
        smh_set_error("smh_sbom_compile_patterns: more than 199 patterns end in one oracle state (the reference's "
                      "state_final_multi rows hold 200 entries)");
        return NULL;
    }
    smh_sbom *sb = sbom_compile(pattern_flat, m, p_size, alphabet, trans, fm, idcounter);
    free(trans); free(fm);
    return sb;
}

int smh_sbom_get_info(const smh_sbom *sb, smh_sbom_info *out)
{
    if (!sb || sb->magic != SMH_MAGIC_SBOM || !out) {
        smh_set_error("smh_sbom_get_info: bad handle");
        return SMH_EINVAL;
    }
    memset(out, 0, sizeof *out);
    out->alphabet = (uint32_t)sb->alphabet;
    out->m = (uint32_t)sb->m;
    out->states = sb->states;
    out->patterns = sb->n_patterns;
    out->listed = sb->listed;
    out->tuned_engine = sb->wm ? SMH_ALGO_WM : SMH_ALGO_AC;
    return SMH_OK;
}

void smh_sbom_free(smh_sbom *sb)
{
    if (!sb || sb->magic != SMH_MAGIC_SBOM) return;
    if (sb->dev) smh_sbom_dev_free(sb->dev);
    sb->dev = NULL;
    smh_sbom_host_free(sb);
}

/* ------------------------------------------------------------------ legacy names */
struct sbom_table *preproc_sbom(unsigned char **pattern, int m, int p_size, int alphabet, int *state_transition,
                                unsigned int *state_final_multi)
{
    if (m < 1 || p_size < 1 || alphabet < 1 || alphabet > 256) fail("preproc_sbom: bad arguments\n");
    struct smh_sbom_table_box *box = (struct smh_sbom_table_box *)calloc(1, sizeof *box);
    unsigned char *flat = (unsigned char *)malloc((size_t)m * p_size);
    if (!box || !flat) fail("Could not initialize table\n");
    uint32_t pc = 0;
    const uint32_t idcounter = sbom_fill_tables(pattern, NULL, m, p_size, alphabet, state_transition, state_final_multi, &pc);
    if (!idcounter) fail("preproc_sbom: more than 199 patterns end in one state (state_final_multi rows hold 200 entries)\n");
    for (int j = 0; j < p_size; ++j) memcpy(flat + (size_t)j * m, pattern[j], (size_t)m);
    box->pub.idcounter = idcounter;
    box->pub.patterncounter = pc;
    box->pub.zerostate = NULL; /* the pointer graph of the reference is not materialised */
    box->magic = SMH_MAGIC_SBOM;
    box->sb = sbom_compile(flat, m, p_size, alphabet, state_transition, state_final_multi, idcounter);
    free(flat);
    if (!box->sb) {
        fputs(smh_last_error(), stderr);
        fail("\npreproc_sbom: could not compile the oracle\n");
    }
    return &box->pub;
}

void free_sbom(struct sbom_table *table, int m)
{
    (void)m;
    if (!table) return;
    struct smh_sbom_table_box *box = (struct smh_sbom_table_box *)table;
    if (box->magic != SMH_MAGIC_SBOM) fail("free_sbom: not a table from preproc_sbom\n");
    smh_sbom_free(box->sb);
    box->magic = 0;
    free(box);
}
