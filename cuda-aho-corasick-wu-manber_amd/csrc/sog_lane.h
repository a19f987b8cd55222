/*
 * csrc/sog_lane.h -- what one lane of the SOG table-walking kernel does.
 *
 * The reference's loop (sog/sog8.c:97-115, cuda/cuda_sog.cu:60-118) over the columns [a, a + span), a column
 * being the first byte of a 3-gram: E = (E << 1) | T8[3-gram]; when bit 5 of E is clear the six 3-grams of
 * an 8-byte window have passed and the window that starts five bytes earlier is verified: hash, 2-level bitmap,
 * binary search over the sorted hashes, 8-byte compare (sog/sog8.c:51-95).  The shift-or state is warmed up over
 * the five columns in front of the span, so every window is tested by exactly one lane.
 */
#ifndef SMH_SOG_LANE_H
#define SMH_SOG_LANE_H

#include "lane_common.h"

#define SMH_SOG_TABLE_SPAN 256u /* columns per lane */

SMH_LANE uint32_t smh_sog_get32(const uint8_t *p) { return ((uint32_t)p[0] << 24) + ((uint32_t)p[1] << 16) + ((uint32_t)p[2] << 8) + (uint32_t)p[3]; }

/* sog_rkbt_verification8 (sog/sog8.c:51-95): 1 when the 8 bytes at w equal a pattern */
SMH_LANE uint32_t smh_sog_verify(const uint8_t *w, const uint32_t *hs_sorted, const int32_t *index, const uint8_t *hs2,
                                 const uint8_t *patterns, int p_size)
{
    const uint32_t hs = smh_sog_get32(w) ^ smh_sog_get32(w + 4);
    const uint32_t lvl = ((hs >> 16) ^ hs) & 0xFFFFu;
    if (!((hs2[lvl >> 3] >> (lvl & 7u)) & 1u)) return 0;
    int lo = 0, hi = p_size - 1;
    while (hi >= lo) {
        int mid = (lo + hi) / 2;
        const uint32_t hp = hs_sorted[mid];
        if (hs > hp) {
            lo = mid + 1;
        } else if (hs < hp) {
            hi = mid - 1;
        } else {
            while (mid > 0 && hs_sorted[mid - 1] == hs) --mid; /* duplicates and patterns with the same hash */
            do {
                const uint8_t *p = patterns + (uint64_t)(uint32_t)index[mid] * 8u;
                int l = 0;
                while (l < 8 && p[l] == w[l]) ++l;
                if (l == 8) return 1;
                ++mid;
            } while (mid < p_size && hs_sorted[mid] == hs);
            return 0;
        }
    }
    return 0;
}

SMH_LANE uint32_t smh_sog_table_thread(uint64_t gthread, uint64_t nthreads, const uint8_t *text, uint64_t n, const uint8_t *t8,
                                       const uint32_t *hs_sorted, const int32_t *index, const uint8_t *hs2,
                                       const uint8_t *patterns, int p_size)
{
    if (n < 8) return 0;
    const uint64_t n_cols = n - 2; /* columns 0 .. n-3 (sog/sog8.c:103) */
    uint32_t cnt = 0;
    for (uint64_t a = gthread * SMH_SOG_TABLE_SPAN; a < n_cols; a += nthreads * SMH_SOG_TABLE_SPAN) {
        uint64_t end = a + SMH_SOG_TABLE_SPAN;
        if (end > n_cols) end = n_cols;
        uint32_t E = 0xffu; /* sog/sog8.c:99 */
        for (uint64_t c = a >= 5 ? a - 5 : 0; c < end; ++c) {
            const uint32_t g = (uint32_t)text[c] + ((uint32_t)text[c + 1] << 8) + ((uint32_t)text[c + 2] << 16);
            E = ((E << 1) | t8[g]) & 0xffu;
            if (c < a || (E & 0x20u)) continue;
            cnt += smh_sog_verify(text + c - 5, hs_sorted, index, hs2, patterns, p_size); /* E has seen six columns: c >= 5 */
        }
    }
    return cnt;
}

#endif
