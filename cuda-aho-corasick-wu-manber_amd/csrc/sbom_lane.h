/*
 * csrc/sbom_lane.h -- what one lane of the SBOM table-walking kernel does.
 *
 * The reference's loop (sbom/sbom.c:128-172, cuda/cuda_sbom.cu:88-118) over the END columns
 * [a, a + span): walk the factor oracle from text[column] backwards for at most m symbols; when all
 * m were read and the state lists patterns, compare them with the window and count the column once
 * if one is equal, then move one column on; otherwise skip max(m - j, 1) columns.  The chain
 * restarts at the first column of every span, as it does per thread in the reference's kernels.
 */
#ifndef SMH_SBOM_LANE_H
#define SMH_SBOM_LANE_H

#include "lane_common.h"

#define SMH_SBOM_TABLE_SPAN 256u /* END columns per lane */

SMH_LANE uint32_t smh_sbom_lane_table(const uint8_t *text, uint64_t n, uint64_t a, uint64_t span,
                                      const int32_t *transition, const uint32_t *final_off,
                                      const uint32_t *final_ids, const uint8_t *patterns, int m, int alphabet)
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
            if (c >= (uint32_t)alphabet) break;
            const int32_t s = transition[(uint64_t)r * (uint32_t)alphabet + c];
            if (s <= 0) break; /* row 0 marks "no edge" with 0, the other rows with -1 */
            r = (uint32_t)s;
            ++j;
        }
        const uint32_t lo = final_off[r], hi = final_off[r + 1];
        if (j == m && hi > lo) {
            const uint8_t *w = text + column - (uint64_t)(m - 1);
            for (uint32_t i = lo; i < hi; ++i) {
                const uint8_t *p = patterns + (uint64_t)final_ids[i] * (uint32_t)m;
                int l = 0;
                while (l < m && p[l] == w[l]) ++l;
                if (l == m) {
                    ++cnt;
                    break;
                }
            }
            ++column;
        } else {
            column += (uint64_t)(m - j > 1 ? m - j : 1);
        }
    }
    return cnt;
}

SMH_LANE uint32_t smh_sbom_table_thread(uint64_t gthread, uint64_t nthreads, const uint8_t *text, uint64_t n,
                                        const int32_t *transition, const uint32_t *final_off,
                                        const uint32_t *final_ids, const uint8_t *patterns, int m, int alphabet)
{
    if (n < (uint64_t)m) return 0;
    uint32_t cnt = 0;
    for (uint64_t a = gthread * SMH_SBOM_TABLE_SPAN; a < n; a += nthreads * SMH_SBOM_TABLE_SPAN)
        cnt += smh_sbom_lane_table(text, n, a, SMH_SBOM_TABLE_SPAN, transition, final_off, final_ids, patterns, m, alphabet);
    return cnt;
}

#endif
