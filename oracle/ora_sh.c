/*
 * oracle/ora_sh.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the reference's Set-Horspool CPU path, sh/sh.c, with the caller conventions of
 * main.c (multish, main.c:158-196).  Same observable results: state numbering of the REVERSED
 * trie, the flat state_transition / state_final tables, idcounter / patterncounter, and the match
 * count.  The trie is read back from the flat table instead of a pointer graph.
 *
 * preBmBc (main.c:173) lives in the reference's missing ../helper.o; ora_pre_bmbc states the textbook
 * set-Horspool bad-character table it stands for (Navarro & Raffinot, "Flexible Pattern Matching in
 * Strings", 3.3.2): bmBc[c] = min over patterns and over i < m-1 with p[i] == c of (m-1-i), else m.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* sh/sh.c:37-63 sh_init + sh/sh.c:82-149 sh_addstring, for every pattern (sh/sh.c:178-196):
 * patterns are inserted BACKWARDS (last symbol first); a new state gets the next id and
 * state_transition[parent*alphabet + symbol] = id; the state reached after m symbols is marked
 * final once (duplicates do not bump patterncounter).  Row 0 is zeroed by sh_init; every other row
 * keeps the caller's -1 (main.c:410-412). */
void ora_preproc_sh(const uint8_t *const *pattern, int m, int p_size, int alphabet,
                    int32_t *state_transition, uint32_t *state_final,
                    uint32_t *idcounter_out, uint32_t *patterncounter_out)
{
    uint32_t idcounter = 1, patterncounter = 0;
    for (int c = 0; c < alphabet; ++c) state_transition[c] = 0;
    for (int i = 0; i < p_size; ++i) {
        const uint8_t *s = pattern[i];
        uint32_t state = 0;
        for (int j = m - 1; j >= 0; --j) {
            const int32_t nx = state_transition[(size_t)state * alphabet + s[j]];
            /* "no edge" reads 0 in row 0 and -1 elsewhere; ids are > 0 */
            if (nx > 0) {
                state = (uint32_t)nx;
            } else {
                state_transition[(size_t)state * alphabet + s[j]] = (int32_t)idcounter;
                state = idcounter++;
            }
        }
        if (!state_final[state]) {
            state_final[state] = 1;
            ++patterncounter;
        }
    }
    if (idcounter_out) *idcounter_out = idcounter;
    if (patterncounter_out) *patterncounter_out = patterncounter;
}

void ora_pre_bmbc(const uint8_t *const *pattern, int m, int p_size, int alphabet, int32_t *bmBc)
{
    for (int c = 0; c < alphabet; ++c) bmBc[c] = m;
    for (int i = 0; i < p_size; ++i)
        for (int j = 0; j < m - 1; ++j)
            if (m - 1 - j < bmBc[pattern[i][j]]) bmBc[pattern[i][j]] = m - 1 - j;
}

/* sh/sh.c:151-176 search_sh over the flat tables: at every visited column walk the reversed trie
 * from text[column] backwards for up to m symbols; count the column when the walk ends in a final
 * state; advance by bmBc[text[column]]. */
uint64_t ora_search_sh(int m, const uint8_t *text, int64_t n, int alphabet,
                       const int32_t *state_transition, const uint32_t *state_final, const int32_t *bmBc)
{
    uint64_t matches = 0;
    int64_t column = m - 1;
    while (column < n) {
        uint32_t r = 0;
        int j = 0;
        while (j < m) {
            const int32_t s = state_transition[(size_t)r * alphabet + text[column - j]];
            if (s <= 0) break;
            r = (uint32_t)s;
            ++j;
        }
        if (state_final[r]) ++matches;
        column += bmBc[text[column]];
    }
    return matches;
}
