/*
 * csrc/corpus.c -- synthetic corpus helper (plain C).
 *
 * Clean-room stand-in for the reference's missing ../helper.c
 * (load_files, create_multiple_pattern_with_hits: main.c:49,453 -- the file and
 * its data sets are not part of the upstream repository).  The stream is the
 * counter-based splitmix64 of SURVEY.md 8c: symbol i = mix(seed + (i+1)*G) % alphabet,
 * so the GPU (corpus_kernels.hip) and the host produce the same text for any
 * slice, and patterns can be sampled from a text that only exists in HBM.
 */
#include "smh_internal.h"

#define SMH_GOLDEN 0x9E3779B97F4A7C15ULL

uint64_t smh_splitmix64_at(uint64_t seed, uint64_t index)
{
    uint64_t z = seed + (index + 1) * SMH_GOLDEN;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void smh_corpus_text_host(unsigned char *out, uint64_t n, uint64_t offset, uint64_t seed, int alphabet)
{
    const uint64_t a = (uint64_t)alphabet;
    for (uint64_t i = 0; i < n; ++i) out[i] = (unsigned char)(smh_splitmix64_at(seed, offset + i) % a);
}

void smh_corpus_patterns(unsigned char *out, int m, int p_size, uint64_t seed, int alphabet,
                         uint64_t text_seed, uint64_t n_text, int from_text_every)
{
    smh_corpus_text_host(out, (uint64_t)m * (uint64_t)p_size, 0, seed, alphabet);
    if (from_text_every <= 0 || n_text < (uint64_t)m) return;
    for (int j = 0; j < p_size; j += from_text_every) {
        uint64_t o = smh_splitmix64_at(seed ^ 0x5DEECE66DULL, (uint64_t)j) % (n_text - (uint64_t)m + 1);
        smh_corpus_text_host(out + (size_t)j * m, (uint64_t)m, o, text_seed, alphabet);
    }
}

void smh_shard_range(uint64_t n, int n_shards, int shard, int m, uint64_t *begin, uint64_t *end)
{
    /* main.c:467-477: [i*c, (i+1)*c + (m-1)) clipped to n, c = ceil(n / R) */
    uint64_t R = n_shards < 1 ? 1u : (uint64_t)n_shards;
    uint64_t c = (n + R - 1) / R;
    uint64_t b = (uint64_t)shard * c;
    uint64_t e = ((uint64_t)shard + 1) * c + (uint64_t)(m > 0 ? m - 1 : 0);
    if (b > n) b = n;
    if (e > n) e = n;
    *begin = b;
    *end = e;
}
