/*
 * oracle/ora_sbom.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the reference's Set Backward Oracle Matching CPU path, sbom/sbom.c, with the caller
 * conventions of main.c (multisbom, main.c:197-231).  Same observable results: state numbering, the
 * flat state_transition (trie edges AND the oracle's external transitions), state_final_multi
 * ({count, pattern ids...} in 200-entry rows), idcounter / patterncounter, and the match count.
 * The pointer graph is replaced by the flat table itself plus a private supply (fail) array.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define NONE 0xFFFFFFFFu

static int has_edge(const int32_t *t, uint32_t state, int alphabet, int c)
{
    /* "no edge" reads 0 in row 0 (sbom_init, sbom/sbom.c:46-49) and -1 elsewhere (main.c:410-412) */
    return t[(size_t)state * alphabet + c] > 0;
}

/* sbom/sbom.c:20-50 sbom_init + :52-126 sbom_addstring for every pattern (:198-214) */
void ora_preproc_sbom(const uint8_t *const *pattern, int m, int p_size, int alphabet,
                      int32_t *state_transition, uint32_t *state_final_multi,
                      uint32_t *idcounter_out, uint32_t *patterncounter_out)
{
    const size_t cap = (size_t)m * p_size + 1;
    uint32_t *fail = (uint32_t *)malloc(cap * sizeof(uint32_t));
    uint32_t idcounter = 1, patterncounter = 0;
    fail[0] = NONE; /* zerostate->fail = NULL */
    for (int c = 0; c < alphabet; ++c) state_transition[c] = 0;
    for (int i = 0; i < p_size; ++i) {
        const uint8_t *s = pattern[i];
        uint32_t state = 0;
        int j = m - 1, done = 0;
        /* follow what exists -- trie edges and external transitions alike (sbom/sbom.c:62-72) */
        while (!done && has_edge(state_transition, state, alphabet, s[j])) {
            state = (uint32_t)state_transition[(size_t)state * alphabet + s[j]];
            if (j <= 0) done = 1;
            j--;
        }
        if (!done) {
            while (j >= 0) {
                const int c = s[j];
                const uint32_t next = idcounter++;
                state_transition[(size_t)state * alphabet + c] = (int32_t)next;
                /* external transitions along the supply chain of the parent (sbom/sbom.c:98-112) */
                uint32_t k = fail[state];
                while (k != NONE && !has_edge(state_transition, k, alphabet, c)) {
                    state_transition[(size_t)k * alphabet + c] = (int32_t)next;
                    k = fail[k];
                }
                fail[next] = k != NONE ? (uint32_t)state_transition[(size_t)k * alphabet + c] : 0u;
                state = next;
                j--;
            }
        }
        /* sbom/sbom.c:114-125: every pattern is appended, duplicates included */
        uint32_t *row = state_final_multi + (size_t)state * 200;
        const uint32_t num = row[0];
        row[0] = num + 1;
        row[num + 1] = patterncounter;
        patterncounter++;
    }
    free(fail);
    if (idcounter_out) *idcounter_out = idcounter;
    if (patterncounter_out) *patterncounter_out = patterncounter;
}

/* sbom/sbom.c:128-172 search_sbom over the flat tables */
uint64_t ora_search_sbom(const uint8_t *pattern_flat, int m, const uint8_t *text, int64_t n, int alphabet,
                         const int32_t *state_transition, const uint32_t *state_final_multi)
{
    uint64_t matches = 0;
    int64_t column = m - 1;
    while (column < n) {
        uint32_t r = 0;
        int j = 0;
        while (j < m && has_edge(state_transition, r, alphabet, text[column - j])) {
            r = (uint32_t)state_transition[(size_t)r * alphabet + text[column - j]];
            j++;
        }
        const uint32_t *row = state_final_multi + (size_t)r * 200;
        if (row[0] > 0 && j == m) {
            for (uint32_t i = 0; i < row[0]; ++i)
                if (memcmp(pattern_flat + (size_t)row[i + 1] * m, text + column - m + 1, (size_t)m) == 0) {
                    matches++;
                    break;
                }
            column++;
        } else {
            column += m - j > 1 ? m - j : 1;
        }
    }
    return matches;
}
