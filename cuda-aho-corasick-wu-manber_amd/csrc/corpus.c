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
#include "corpus_gen.h"
#include <string.h>

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

/* ---- the non-uniform kinds (corpus_gen.h) ---- */
int smh_corpus_tabs_build(struct smh_corpus_tabs *T, uint64_t seed, int alphabet, int kind)
{
    memset(T, 0, sizeof *T);
    if (alphabet < 1 || alphabet > 256) { smh_set_error("corpus: alphabet %d", alphabet); return SMH_EINVAL; }
    if (kind == SMH_CORPUS_DNA_REPEATS) {
        if (alphabet != 4) { smh_set_error("corpus: SMH_CORPUS_DNA_REPEATS is a 4-letter text"); return SMH_EINVAL; }
        static const uint32_t w[4] = {32768u, 16384u, 9830u, 6554u}; /* 0.50 / 0.25 / 0.15 / 0.10 */
        for (uint32_t c = 0; c < 64; ++c) {
            /* the context picks which symbol gets which probability: one of the 24 orders */
            uint32_t perm[4] = {0, 1, 2, 3}, k = (uint32_t)(smh_splitmix64_at(0xD1A5EEDULL, c) % 24u);
            for (uint32_t i = 0; i < 3; ++i) {
                const uint32_t left = 4u - i, pick = k % left;
                k /= left;
                const uint32_t t = perm[i]; perm[i] = perm[i + pick]; perm[i + pick] = t;
            }
            uint32_t prob[4];
            for (uint32_t i = 0; i < 4; ++i) prob[perm[i]] = w[i];
            T->markov[c][0] = (uint16_t)prob[0];
            T->markov[c][1] = (uint16_t)(prob[0] + prob[1]);
            T->markov[c][2] = (uint16_t)(prob[0] + prob[1] + prob[2] > 65535u ? 65535u : prob[0] + prob[1] + prob[2]);
        }
    } else if (kind == SMH_CORPUS_SKEWED) {
        uint64_t W[256], total = 0;
        for (int k = 0; k < alphabet; ++k) {
            const uint64_t d = (uint64_t)k + 4u;
            W[k] = ((uint64_t)1 << 40) / (alphabet > 32 ? d * d : d);
            total += W[k];
        }
        uint64_t cum = W[0];
        int k = 0;
        for (uint32_t i = 0; i < 1024u; ++i) {
            while (k + 1 < alphabet && cum * 1024u <= (uint64_t)i * total) cum += W[++k];
            T->quant[i] = (uint8_t)k;
        }
    } else if (kind == SMH_CORPUS_PLANTED) {
        for (uint32_t j = 0; j < SMH_CORPUS_WORD; ++j) T->word[j] = (uint8_t)(smh_splitmix64_at(seed ^ 0x3A27EDULL, j) % (uint64_t)alphabet);
    } else {
        smh_set_error("corpus: unknown kind %d", kind);
        return SMH_EINVAL;
    }
    return SMH_OK;
}

static void text_kind_with(const struct smh_corpus_tabs *T, unsigned char *out, uint64_t n, uint64_t offset, uint64_t seed,
                           int alphabet, int kind)
{
    const uint64_t end = offset + n;
    for (uint64_t b = offset / SMH_CORPUS_BLOCK; b * SMH_CORPUS_BLOCK < end; ++b) {
        struct smh_cg_block B;
        smh_cg_block_of(seed, b, kind, T, &B);
        uint32_t ctx = 0;
        uint8_t unit[32];
        for (uint32_t p = 0; p < SMH_CORPUS_BLOCK; ++p) {
            const uint64_t i = b * SMH_CORPUS_BLOCK + p;
            if (i >= end) break;
            const uint32_t sym = smh_cg_next(T, &B, kind, (uint32_t)alphabet, b, p, &ctx, unit);
            if (i >= offset) out[i - offset] = (unsigned char)sym;
        }
    }
}

int smh_corpus_text_host_kind(unsigned char *out, uint64_t n, uint64_t offset, uint64_t seed, int alphabet, int kind)
{
    if (kind == SMH_CORPUS_UNIFORM) {
        if (alphabet < 1 || alphabet > 256) { smh_set_error("corpus: alphabet %d", alphabet); return SMH_EINVAL; }
        smh_corpus_text_host(out, n, offset, seed, alphabet);
        return SMH_OK;
    }
    struct smh_corpus_tabs T;
    const int rc = smh_corpus_tabs_build(&T, seed, alphabet, kind);
    if (rc != SMH_OK) return rc;
    text_kind_with(&T, out, n, offset, seed, alphabet, kind);
    return SMH_OK;
}

int smh_corpus_patterns_kind(unsigned char *out, int m, int p_size, uint64_t seed, int alphabet, uint64_t text_seed,
                             uint64_t n_text, int from_text_every, int kind)
{
    if (kind == SMH_CORPUS_UNIFORM) {
        smh_corpus_patterns(out, m, p_size, seed, alphabet, text_seed, n_text, from_text_every);
        return SMH_OK;
    }
    if (m < 1 || p_size < 1) { smh_set_error("corpus: bad pattern shape"); return SMH_EINVAL; }
    struct smh_corpus_tabs Tt, To; /* of the text, and of the unrelated text the other patterns are slices of */
    const uint64_t other = seed ^ 0x9A77E2ULL;
    int rc = smh_corpus_tabs_build(&Tt, text_seed, alphabet, kind);
    if (rc == SMH_OK) rc = smh_corpus_tabs_build(&To, other, alphabet, kind);
    if (rc != SMH_OK) return rc;
    /* patterns that are not substrings of the text have the text's statistics all the same */
    for (int j = 0; j < p_size; ++j) {
        if (from_text_every > 0 && n_text >= (uint64_t)m && j % from_text_every == 0) {
            const uint64_t o = smh_splitmix64_at(seed ^ 0x5DEECE66DULL, (uint64_t)j) % (n_text - (uint64_t)m + 1);
            text_kind_with(&Tt, out + (size_t)j * m, (uint64_t)m, o, text_seed, alphabet, kind);
        } else {
            text_kind_with(&To, out + (size_t)j * m, (uint64_t)m, (uint64_t)j * 4099u + 7u, other, alphabet, kind);
        }
    }
    if (kind == SMH_CORPUS_PLANTED && (uint32_t)m <= SMH_CORPUS_WORD) memcpy(out, Tt.word, (size_t)m); /* pattern 0: the planted word's first m symbols */
    return SMH_OK;
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
