/*
 * oracle/ora_sog.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the reference's SOG path for 8-byte patterns, sog/sog8.c: 3-gram bit table T8 (bit i of T8[g]
 * cleared when g is the 3-gram at offset i of a pattern, sog/sog8.c:138-145), pattern hashes GET32 ^ GET32+4
 * sorted by the reference's quicksort (sog/sog8.c:30-49) together with their permutation, 2-level bitmap, and
 * the shift-or search loop with hash / binary-search / memcmp verification (sog/sog8.c:51-115).
 *
 * ONE deliberate difference: sog_add_pattern computes the 2-level index from `hs` BEFORE assigning it
 * (sog/sog8.c:124,135 -- an uninitialised read), so the reference's bitmap, and with it its match count, is not
 * a function of its inputs.  Here the bitmap bit is the one the search tests (from the pattern's real hash),
 * which makes the count the definition: the number of 8-byte windows that equal a pattern.  T8, the sorted
 * hashes and the permutation ARE deterministic in the reference and are pinned against it.
 */
#include "oracle.h"
#include <string.h>

#define GET32(a) (((uint32_t)(a)[0] << 24) + ((uint32_t)(a)[1] << 16) + ((uint32_t)(a)[2] << 8) + (uint32_t)(a)[3])
#define GET3GRAM(a) ((uint32_t)(a)[0] + ((uint32_t)(a)[1] << 8) + ((uint32_t)(a)[2] << 16))

/* sog/sog8.c:30-49 */
static void ora_sog_sort(uint32_t *hs, int32_t *index, int beg, int end)
{
    if (end > beg + 1) {
        uint32_t piv = hs[beg];
        int l = beg + 1, r = end;
        while (l < r) {
            if (hs[l] <= piv) {
                l++;
            } else {
                --r;
                uint32_t th = hs[l]; hs[l] = hs[r]; hs[r] = th;
                int32_t ti = index[l]; index[l] = index[r]; index[r] = ti;
            }
        }
        --l;
        uint32_t th = hs[l]; hs[l] = hs[beg]; hs[beg] = th;
        int32_t ti = index[l]; index[l] = index[beg]; index[beg] = ti;
        ora_sog_sort(hs, index, beg, l);
        ora_sog_sort(hs, index, r, end);
    }
}

/* sog/sog8.c:117-175 (preproc_sog8, sog_reset_patterns, sog_add_pattern) */
void ora_preproc_sog8(uint8_t *T8, uint32_t *scanner_hs, int32_t *scanner_index, uint8_t *scanner_hs2,
                      const uint8_t *const *pattern, int p_size)
{
    memset(T8, 0xff, (size_t)1 << 24);
    memset(scanner_hs2, 0, 32 * 256);
    for (int j = 0; j < p_size; ++j) {
        const uint8_t *p = pattern[j];
        scanner_index[j] = j;
        scanner_hs[j] = GET32(p) ^ GET32(p + 4);
        const uint32_t hs = scanner_hs[j];
        const uint16_t hs2level = (uint16_t)((hs >> 16) ^ hs); /* defined form of sog/sog8.c:135 */
        scanner_hs2[hs2level >> 3] |= (uint8_t)(1u << (hs2level & 7u));
        for (unsigned i = 0; i < 6; ++i) T8[GET3GRAM(p + i)] &= (uint8_t)(0xffu - (1u << i));
    }
    ora_sog_sort(scanner_hs, scanner_index, 0, p_size);
}

/* sog/sog8.c:51-95 */
static int ora_sog_verify(const uint32_t *scanner_hs, const int32_t *scanner_index, const uint8_t *scanner_hs2,
                          const uint8_t *const *pattern, const uint8_t *text, int p_size)
{
    uint32_t hs = GET32(text) ^ GET32(text + 4);
    uint16_t hs2level = (uint16_t)((hs >> 16) ^ hs);
    if (scanner_hs2[hs2level >> 3] & (1u << (hs2level & 7u))) {
        int lo = 0, hi = p_size - 1;
        while (hi >= lo) {
            int mid = (lo + hi) / 2;
            uint32_t hs_pat = scanner_hs[mid];
            if (hs > hs_pat) {
                lo = ++mid;
            } else if (hs < hs_pat) {
                hi = --mid;
            } else {
                while (mid > 0 && hs == scanner_hs[mid - 1]) mid--;
                do {
                    if (memcmp(text, pattern[scanner_index[mid]], 8) == 0) return 1;
                    mid++;
                } while (mid < p_size && hs == scanner_hs[mid]);
                break;
            }
        }
    }
    return -1;
}

/* sog/sog8.c:97-115 search_sog8 (m = 8, B = 3: the window starts at column - m + B = column - 5) */
uint64_t ora_search_sog8(const uint8_t *T8, const uint32_t *scanner_hs, const int32_t *scanner_index,
                         const uint8_t *scanner_hs2, const uint8_t *const *pattern, const uint8_t *text, int64_t n,
                         int p_size)
{
    uint8_t E = 0xff;
    uint64_t matches = 0;
    for (int64_t column = 0; column < n - 2; column++) {
        E = (uint8_t)((E << 1) | T8[GET3GRAM(text + column)]);
        if (E & 0x20) continue;
        if (ora_sog_verify(scanner_hs, scanner_index, scanner_hs2, pattern, text + column - 5, p_size) != -1) matches++;
    }
    return matches;
}
