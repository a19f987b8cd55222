/*
 * csrc/sog_host.c -- host side of SOG, the reference's shift-or with 3-grams for 8-byte patterns
 * (sog/sog8.c, smatcher.h:75-80,108-109), third sibling algorithm behind the same API.
 *
 *   preproc_sog8   fills the caller's tables as sog/sog8.c:117-175 does -- T8 (2^24 bytes: bit i of T8[g] CLEARED
 *                  when g is the 3-gram at offset i of some pattern), scanner_hs (hash of every pattern, sorted
 *                  ascending by the reference's own quicksort, sog/sog8.c:30-49), scanner_index (the permutation
 *                  that sort applies) -- bit-identically, and scanner_hs2 with DEFINED contents: the reference
 *                  sets that 2-level bitmap from an uninitialised variable (sog/sog8.c:124,135: `hs` is read
 *                  before it is assigned), so which windows its search then drops depends on stack contents.
 *                  Here the bit is the one search_sog8 tests: hs2level = (uint16)((hs >> 16) ^ hs) of the
 *                  pattern's real hash (sog/sog8.c:54-57).
 *   search_sog8    the count the algorithm is meant to return: the number of 8-byte windows of the text that
 *                  equal a pattern (every column whose six 3-grams pass and whose window verifies,
 *                  sog/sog8.c:97-115) -- the same quantity search_ac / search_wu return for m = 8.  Computed on
 *                  the GPU (smh_runtime.hip).
 *   smh_sog_*      handles over the caller's tables.
 */
#include "smh_internal.h"

#include <stdlib.h>
#include <string.h>

#define SOG_GET32(a) (((uint32_t)(a)[0] << 24) + ((uint32_t)(a)[1] << 16) + ((uint32_t)(a)[2] << 8) + (uint32_t)(a)[3])
#define SOG_GET3GRAM(a) ((uint32_t)(a)[0] + ((uint32_t)(a)[1] << 8) + ((uint32_t)(a)[2] << 16))

/* sog/sog8.c:30-49 my_sort: quicksort on (hash, index) pairs, pivot = first element, elements <= pivot left.
 * Restated with the same partition order so that equal hashes end up in the reference's order. */
static void sog_sort(uint32_t *hs, int *index, int beg, int end)
{
    while (end > beg + 1) {
        const uint32_t piv = hs[beg];
        int l = beg + 1, r = end;
        while (l < r) {
            if (hs[l] <= piv) {
                ++l;
            } else {
                --r;
                const uint32_t th = hs[l]; hs[l] = hs[r]; hs[r] = th;
                const int ti = index[l]; index[l] = index[r]; index[r] = ti;
            }
        }
        --l;
        { const uint32_t th = hs[l]; hs[l] = hs[beg]; hs[beg] = th; }
        { const int ti = index[l]; index[l] = index[beg]; index[beg] = ti; }
        sog_sort(hs, index, beg, l); /* the reference recurses on both halves; the right one is the loop here */
        beg = r;
    }
}

void preproc_sog8(uint8_t *T8, uint32_t *scanner_hs, int *scanner_index, uint8_t *scanner_hs2, unsigned char **pattern,
                  int m, unsigned char *text, int n, int p_size, int B)
{
    (void)text; (void)n; (void)B;
    if (m != 8) fail("preproc_sog8: SOG is built for patterns of length 8 (sog/sog8.c:117-146)\n");
    if (!T8 || !scanner_hs || !scanner_index || !scanner_hs2 || !pattern || p_size < 1) fail("preproc_sog8: bad arguments\n");
    memset(T8, 0xff, (size_t)1 << 24);   /* sog_reset_patterns, sog/sog8.c:148-159 */
    memset(scanner_hs2, 0, 32 * 256);
    for (int j = 0; j < p_size; ++j) {
        const unsigned char *p = pattern[j];
        const uint32_t hs = SOG_GET32(p) ^ SOG_GET32(p + 4);
        scanner_index[j] = j;
        scanner_hs[j] = hs;
        const uint16_t hs2level = (uint16_t)((hs >> 16) ^ hs); /* the value search tests, sog/sog8.c:54-57 */
        scanner_hs2[hs2level >> 3] |= (uint8_t)(1u << (hs2level & 7u));
        for (unsigned i = 0; i < 6; ++i) T8[SOG_GET3GRAM(p + i)] &= (uint8_t)(0xffu - (1u << i));
    }
    sog_sort(scanner_hs, scanner_index, 0, p_size);
}

void smh_sog_free(struct smh_sog *sg)
{
    if (!sg) return;
    if (sg->dev) smh_sog_dev_free(sg->dev);
    smh_wm_free(sg->wm);
    free(sg->t8);
    free(sg->hs);
    free(sg->index);
    free(sg->hs2);
    free(sg->patterns);
    sg->magic = 0;
    free(sg);
}

struct smh_sog *smh_sog_compile_tables(const uint8_t *T8, const uint32_t *scanner_hs, const int *scanner_index,
                                       const uint8_t *scanner_hs2, const unsigned char *pattern_flat, int p_size)
{
    if (!T8 || !scanner_hs || !scanner_index || !scanner_hs2 || !pattern_flat || p_size < 1) {
        smh_set_error("smh_sog_compile_tables: bad arguments");
        return NULL;
    }
    for (int j = 0; j < p_size; ++j)
        if (scanner_index[j] < 0 || scanner_index[j] >= p_size) {
            smh_set_error("smh_sog_compile_tables: scanner_index[%d] = %d out of range", j, scanner_index[j]);
            return NULL;
        }
    struct smh_sog *sg = (struct smh_sog *)calloc(1, sizeof *sg);
    if (!sg) goto oom;
    sg->magic = SMH_MAGIC_SOG;
    sg->n_patterns = (uint32_t)p_size;
    sg->t8 = (uint8_t *)malloc((size_t)1 << 24);
    sg->hs = (uint32_t *)malloc((size_t)p_size * sizeof(uint32_t));
    sg->index = (int32_t *)malloc((size_t)p_size * sizeof(int32_t));
    sg->hs2 = (uint8_t *)malloc(8192);
    sg->patterns = (unsigned char *)malloc((size_t)p_size * 8);
    if (!sg->t8 || !sg->hs || !sg->index || !sg->hs2 || !sg->patterns) goto oom;
    memcpy(sg->t8, T8, (size_t)1 << 24);
    memcpy(sg->hs, scanner_hs, (size_t)p_size * sizeof(uint32_t));
    memcpy(sg->index, scanner_index, (size_t)p_size * sizeof(int32_t));
    memcpy(sg->hs2, scanner_hs2, 8192);
    memcpy(sg->patterns, pattern_flat, (size_t)p_size * 8);
    /* tuned engine: 8-byte patterns over the byte alphabet are a Wu-Manber set (same count) */
    sg->wm = smh_wm_compile(pattern_flat, 8, p_size, 256);
    if (!sg->wm) { smh_sog_free(sg); return NULL; }
    return sg;
oom:
    smh_set_error("smh_sog_compile_tables: out of memory");
    smh_sog_free(sg);
    return NULL;
}
