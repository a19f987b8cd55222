/*
 * oracle/ora_gen.c -- TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Synthetic corpus used by the parity tests.  The reference ships no data
 * files (main.c:35-116 names ../data-cuda-multi/... which are absent), so the
 * corpus is the counter-based splitmix64 stream SURVEY.md 8c pinned its
 * known-answer counts on:  s_i = seed + (i+1)*0x9E3779B97F4A7C15,
 * z = mix(s_i), symbol = z % sigma.  Being counter-based, any slice of the
 * text can be regenerated independently (used for shards and for sampling
 * patterns out of a text that only lives in GPU memory).
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

#define ORA_GOLDEN 0x9E3779B97F4A7C15ULL

uint64_t ora_splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * ORA_GOLDEN;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void ora_gen_text(uint8_t *out, uint64_t n, uint64_t offset, uint64_t seed, int sigma)
{
    for (uint64_t i = 0; i < n; ++i)
        out[i] = (uint8_t)(ora_splitmix64_at(seed, offset + i) % (uint64_t)sigma);
}

void ora_gen_patterns_uniform(uint8_t *out, int m, int p, uint64_t seed, int sigma)
{
    ora_gen_text(out, (uint64_t)m * (uint64_t)p, 0, seed, sigma);
}

void ora_gen_patterns_mixed(uint8_t *out, int m, int p, uint64_t seed, int sigma,
                            uint64_t text_seed, uint64_t n_text, int from_text_every)
{
    ora_gen_patterns_uniform(out, m, p, seed, sigma);
    if (from_text_every <= 0 || n_text < (uint64_t)m)
        return;
    for (int j = 0; j < p; j += from_text_every) {
        uint64_t o = ora_splitmix64_at(seed ^ 0x5DEECE66DULL, (uint64_t)j) % (n_text - (uint64_t)m + 1);
        ora_gen_text(out + (size_t)j * m, (uint64_t)m, o, text_seed, sigma);
    }
}

void ora_shard_range(int64_t n, int R, int i, int m, int64_t *begin, int64_t *end)
{
    /* main.c:467-477: start = i*ceil(n/R); stop = (i+1)*ceil(n/R) + (m-1), clipped to n */
    int64_t c = (n + R - 1) / R;
    int64_t b = (int64_t)i * c;
    int64_t e = (int64_t)(i + 1) * c + (m - 1);
    if (b > n) b = n;
    if (e > n) e = n;
    *begin = b;
    *end = e;
}

uint64_t ora_fnv1a64(const void *buf, size_t bytes)
{
    const uint8_t *p = (const uint8_t *)buf;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < bytes; ++i) {
        h ^= p[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}

/* ---- definition-level brute force: sort the pattern set, binary-search every window ---- */
static int g_cmp_m;
static int cmp_pat(const void *a, const void *b) { return memcmp(a, b, (size_t)g_cmp_m); }

static int in_set(const uint8_t *sorted, int p, int m, const uint8_t *w)
{
    int lo = 0, hi = p - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        int c = memcmp(sorted + (size_t)mid * m, w, (size_t)m);
        if (c == 0) return 1;
        if (c < 0) lo = mid + 1; else hi = mid - 1;
    }
    return 0;
}

uint64_t ora_positions_bruteforce(const uint8_t *pattern_flat, int m, int p_size,
                                  const uint8_t *text, int64_t n, int64_t *out, uint64_t cap)
{
    if (p_size <= 0 || m <= 0 || n < m) return 0;
    uint8_t *sorted = (uint8_t *)malloc((size_t)p_size * m);
    memcpy(sorted, pattern_flat, (size_t)p_size * m);
    g_cmp_m = m;
    qsort(sorted, (size_t)p_size, (size_t)m, cmp_pat);
    uint64_t found = 0;
    for (int64_t e = m - 1; e < n; ++e) {
        if (in_set(sorted, p_size, m, text + e - m + 1)) {
            if (out && found < cap) out[found] = e;
            ++found;
        }
    }
    free(sorted);
    return found;
}

uint64_t ora_count_bruteforce(const uint8_t *pattern_flat, int m, int p_size,
                              const uint8_t *text, int64_t n)
{
    return ora_positions_bruteforce(pattern_flat, m, p_size, text, n, NULL, 0);
}
