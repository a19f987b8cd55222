/*
 * oracle/ora_wu.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Restatement of the reference's Wu-Manber CPU path, wu/wu.c.  The block
 * hash is the reference's fixed 3-symbol hash ((c0 << nbits) + c1 << nbits) + c2
 * with nbits = m_nBitsInShift = 2 (main.c:431), the prefix hash is
 * (p0 << nbits) + p1 (wu/wu.c:136-138), and the PREFIX tables are the dense
 * [shiftsize x p_size] arrays main.c:436-439 allocates.
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

/* wu/wu.c:18-47 wu_determine_shiftsize: (alphabet-1)*21 + 1 for the listed alphabets */
uint32_t ora_wu_determine_shiftsize(int alphabet)
{
    switch (alphabet) {
    case 2: return 22;
    case 4: return 64;
    case 8: return 148;
    case 20: return 400;
    case 128: return 2668;
    case 256: return 5356;
    case 512: return 10732;
    case 1024: return 21484;
    default: return 0; /* reference: fail("The alphabet size is not supported by wu-manber") */
    }
}

static inline uint32_t block_hash(const uint8_t *s, int nbits)
{
    /* wu/wu.c:63-67 (search) and :121-125 (preproc): three symbols ending at s[0] */
    uint32_t h = s[-2];
    h <<= nbits;
    h += s[-1];
    h <<= nbits;
    h += s[0];
    return h;
}

static inline uint32_t prefix_hash(const uint8_t *s, int nbits)
{
    /* wu/wu.c:75-77 and :136-138: first two symbols of the window / pattern */
    uint32_t h = s[0];
    h <<= nbits;
    h += s[1];
    return h;
}

/* wu/wu.c:211-251 preproc_wu2 (flat patterns).  For q = m..B the block ending
 * at pattern offset q-1 lowers SHIFT[hash] to m-q; blocks with m-q == 0 (the
 * pattern's suffix block) append {prefix hash, pattern index} to bucket `hash`. */
void ora_preproc_wu2(const uint8_t *pattern_flat, int m, int p_size, int alphabet, int B,
                     int nbits, int32_t *SHIFT, int32_t *PREFIX_value, int32_t *PREFIX_index,
                     int32_t *PREFIX_size)
{
    (void)alphabet; /* unused by the reference too */
    for (int j = 0; j < p_size; ++j) {
        const uint8_t *P = pattern_flat + (size_t)j * m;
        for (int q = m; q >= B; --q) {
            uint32_t hash = block_hash(P + q - 1, nbits);
            int32_t shiftlen = m - q;
            if (shiftlen < SHIFT[hash]) SHIFT[hash] = shiftlen;
            if (shiftlen == 0) {
                size_t slot = (size_t)hash * p_size + (size_t)PREFIX_size[hash];
                PREFIX_value[slot] = (int32_t)prefix_hash(P, nbits);
                PREFIX_index[slot] = j;
                PREFIX_size[hash]++;
            }
        }
    }
}

/* wu/wu.c:109-149 preproc_wu (pointer-per-pattern form, otherwise identical) */
void ora_preproc_wu(const uint8_t *const *pattern, int m, int p_size, int alphabet, int B,
                    int nbits, int32_t *SHIFT, int32_t *PREFIX_value, int32_t *PREFIX_index,
                    int32_t *PREFIX_size)
{
    (void)alphabet;
    for (int j = 0; j < p_size; ++j) {
        const uint8_t *P = pattern[j];
        for (int q = m; q >= B; --q) {
            uint32_t hash = block_hash(P + q - 1, nbits);
            int32_t shiftlen = m - q;
            if (shiftlen < SHIFT[hash]) SHIFT[hash] = shiftlen;
            if (shiftlen == 0) {
                size_t slot = (size_t)hash * p_size + (size_t)PREFIX_size[hash];
                PREFIX_value[slot] = (int32_t)prefix_hash(P, nbits);
                PREFIX_index[slot] = j;
                PREFIX_size[hash]++;
            }
        }
    }
}

/* wu/wu.c:151-209 search_wu2: classic skip loop over end columns.  A column
 * with SHIFT == 0 scans its bucket; the first entry whose prefix hash and full
 * m bytes agree counts ONE match for that column (matches++; break;
 * wu/wu.c:91-95) and the window advances by one. */
uint64_t ora_search_wu2(const uint8_t *pattern_flat, int m, int p_size, const uint8_t *text,
                        int64_t n, int nbits, const int32_t *SHIFT, const int32_t *PREFIX_value,
                        const int32_t *PREFIX_index, const int32_t *PREFIX_size)
{
    uint64_t matches = 0;
    int64_t column = m - 1;
    while (column < n) {
        uint32_t hash1 = block_hash(text + column, nbits);
        int32_t shift = SHIFT[hash1];
        if (shift == 0) {
            uint32_t hash2 = prefix_hash(text + column - m + 1, nbits);
            for (int32_t i = 0; i < PREFIX_size[hash1]; ++i) {
                size_t slot = (size_t)hash1 * p_size + (size_t)i;
                if ((int32_t)hash2 == PREFIX_value[slot] &&
                    memcmp(pattern_flat + (size_t)PREFIX_index[slot] * m,
                           text + column - m + 1, (size_t)m) == 0) {
                    ++matches;
                    break;
                }
            }
            ++column;
        } else {
            column += shift;
        }
    }
    return matches;
}

/* wu/wu.c:49-107 search_wu (pointer-per-pattern form) */
uint64_t ora_search_wu(const uint8_t *const *pattern, int m, int p_size, const uint8_t *text,
                       int64_t n, int nbits, const int32_t *SHIFT, const int32_t *PREFIX_value,
                       const int32_t *PREFIX_index, const int32_t *PREFIX_size)
{
    uint64_t matches = 0;
    int64_t column = m - 1;
    while (column < n) {
        uint32_t hash1 = block_hash(text + column, nbits);
        int32_t shift = SHIFT[hash1];
        if (shift == 0) {
            uint32_t hash2 = prefix_hash(text + column - m + 1, nbits);
            for (int32_t i = 0; i < PREFIX_size[hash1]; ++i) {
                size_t slot = (size_t)hash1 * p_size + (size_t)i;
                if ((int32_t)hash2 == PREFIX_value[slot] &&
                    memcmp(pattern[PREFIX_index[slot]], text + column - m + 1, (size_t)m) == 0) {
                    ++matches;
                    break;
                }
            }
            ++column;
        } else {
            column += shift;
        }
    }
    return matches;
}

/* ------------------------------------------------------------------ CSR form
 * The dense [shiftsize x p_size] PREFIX tables of main.c:436-439 are 2 x 2.1 GB at alphabet 256 / 100 000
 * patterns.  The same preprocessing into compressed rows: bucket `hash` holds its entries in the order
 * preproc_wu2 appends them (ascending pattern index, wu/wu.c:231-247), at bucket_off[hash] ..
 * bucket_off[hash + 1].  ora_search_wu_csr is search_wu2 (wu/wu.c:151-209) reading those rows.  Checked
 * equal to the dense restatement and to the compiled reference on the golden vectors (tests/test_oracle.py).
 * SHIFT must be pre-filled with m - B + 1 as for the dense form; bucket_val / bucket_idx hold p_size
 * entries (every pattern lands in exactly one bucket: its suffix block's). */
void ora_preproc_wu_csr(const uint8_t *pattern_flat, int m, int p_size, int B, int nbits, uint32_t shiftsize,
                        int32_t *SHIFT, uint32_t *bucket_off, int32_t *bucket_val, int32_t *bucket_idx)
{
    memset(bucket_off, 0, ((size_t)shiftsize + 1) * sizeof(uint32_t));
    for (int j = 0; j < p_size; ++j) {
        const uint8_t *P = pattern_flat + (size_t)j * m;
        for (int q = m; q >= B; --q) {
            uint32_t hash = block_hash(P + q - 1, nbits);
            int32_t shiftlen = m - q;
            if (shiftlen < SHIFT[hash]) SHIFT[hash] = shiftlen;
            if (shiftlen == 0) bucket_off[hash + 1]++;
        }
    }
    for (uint32_t h = 0; h < shiftsize; ++h) bucket_off[h + 1] += bucket_off[h];
    /* second pass in pattern order, so that every row keeps wu/wu.c's append order; the row cursors
     * live in a scratch copy of the offsets */
    uint32_t *cursor = (uint32_t *)malloc(((size_t)shiftsize + 1) * sizeof(uint32_t));
    memcpy(cursor, bucket_off, ((size_t)shiftsize + 1) * sizeof(uint32_t));
    for (int j = 0; j < p_size; ++j) {
        const uint8_t *P = pattern_flat + (size_t)j * m;
        const uint32_t k = cursor[block_hash(P + m - 1, nbits)]++;
        bucket_val[k] = (int32_t)prefix_hash(P, nbits);
        bucket_idx[k] = j;
    }
    free(cursor);
}

uint64_t ora_search_wu_csr(const uint8_t *pattern_flat, int m, const uint8_t *text, int64_t n, int nbits,
                           const int32_t *SHIFT, const uint32_t *bucket_off, const int32_t *bucket_val,
                           const int32_t *bucket_idx)
{
    uint64_t matches = 0;
    int64_t column = m - 1;
    while (column < n) {
        uint32_t hash1 = block_hash(text + column, nbits);
        int32_t shift = SHIFT[hash1];
        if (shift == 0) {
            uint32_t hash2 = prefix_hash(text + column - m + 1, nbits);
            for (uint32_t k = bucket_off[hash1]; k < bucket_off[hash1 + 1]; ++k) {
                if ((int32_t)hash2 == bucket_val[k] &&
                    memcmp(pattern_flat + (size_t)bucket_idx[k] * m, text + column - m + 1, (size_t)m) == 0) {
                    ++matches;
                    break;
                }
            }
            ++column;
        } else {
            column += shift;
        }
    }
    return matches;
}
