/*
 * csrc/sh_lane.h -- what one lane of the Set-Horspool table-walking kernel does.
 *
 * The reference's loop (sh/sh.c:151-176, cuda/cuda_sh.cu:82-104) over the END columns
 * [a, a + span): walk the reversed trie from text[column] backwards for at most m symbols, count the
 * column when the walk ends in a final state, advance by bmBc[text[column]].  The skip chain restarts
 * at the first column of every span (as it restarts per thread in the reference's kernels and per
 * rank in main.c:467-477); with a valid bad-character table the set of counted columns does not
 * depend on where a chain starts.
 */
#ifndef SMH_SH_LANE_H
#define SMH_SH_LANE_H

#include "lane_common.h"

#define SMH_SH_TABLE_SPAN 256u /* END columns per lane */

template <typename BMBC_T>
SMH_LANE uint32_t smh_sh_lane_table(const uint8_t *text, uint64_t n, uint64_t a, uint64_t span,
                                    const int32_t *transition, const uint32_t *final_, const BMBC_T *bmbc,
                                    int m, int alphabet)
{
    uint64_t end = a + span;
    if (end > n) end = n;
    uint64_t column = a;
    if (column < (uint64_t)(m - 1)) column = (uint64_t)(m - 1);
    uint32_t cnt = 0;
    while (column < end) {
        uint32_t r = 0;
        int j = 0;
        while (j < m) {
            const uint32_t c = text[column - (uint64_t)j];
            if (c >= (uint32_t)alphabet) break; /* a byte outside the alphabet has no edge */
            const int32_t s = transition[(uint64_t)r * (uint32_t)alphabet + c];
            if (s <= 0) break; /* row 0 marks "no edge" with 0, the other rows with -1 */
            r = (uint32_t)s;
            ++j;
        }
        cnt += final_[r] != 0u;
        const uint32_t c0 = text[column];
        const int32_t shift = c0 < (uint32_t)alphabet ? (int32_t)bmbc[c0] : 1;
        column += (uint64_t)(shift < 1 ? 1 : shift);
    }
    return cnt;
}

template <typename BMBC_T>
SMH_LANE uint32_t smh_sh_table_thread(uint64_t gthread, uint64_t nthreads, const uint8_t *text, uint64_t n,
                                      const int32_t *transition, const uint32_t *final_, const BMBC_T *bmbc,
                                      int m, int alphabet)
{
    if (n < (uint64_t)m) return 0;
    uint32_t cnt = 0;
    for (uint64_t a = gthread * SMH_SH_TABLE_SPAN; a < n; a += nthreads * SMH_SH_TABLE_SPAN)
        cnt += smh_sh_lane_table<BMBC_T>(text, n, a, SMH_SH_TABLE_SPAN, transition, final_, bmbc, m, alphabet);
    return cnt;
}

#endif
