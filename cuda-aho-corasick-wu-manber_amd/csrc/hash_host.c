/*
 * csrc/hash_host.c -- host builder of the window-hash engine (hash_engine.h): the Bloom filter of the patterns' rolling hashes
 * (LDS image) and the two-table cuckoo hash of the patterns themselves (device memory), for the distinct patterns of one length.
 * Replaces, for the sets it takes, wu/wu.c:109-149's SHIFT / PREFIX tables and the memcmp of wu/wu.c:88.
 */
#include <stdlib.h>
#include <string.h>
#include "smh_internal.h"
#include "hash_engine.h"

static uint32_t hash_tag(const unsigned char *p, int m) /* == wm_host.c smh_wm_tag == wm_lane.h smh_wm_tag_dwords */
{
    uint32_t h = 0x811C9DC5u;
    for (int j = 0; j < (m + 3) / 4; ++j) {
        uint32_t v = 0;
        for (int b = 0; b < 4 && 4 * j + b < m; ++b) v |= (uint32_t)p[4 * j + b] << (8 * b);
        h = (h ^ v) * 0x9E3779B1u;
        h ^= h >> 15;
    }
    return h;
}

static uint32_t hash_roll(const unsigned char *p, int m)
{
    uint32_t h = 0;
    for (int i = 0; i < m; ++i) h = smh_hash_in(h, p[i]);
    return h & 0xFFFFFFu;
}

void smh_hash_free(struct smh_hashes *k)
{
    if (!k) return;
    smh_hash_dev_free(k->dev);
    free(k->bloom);
    free(k->table);
    free(k);
}

/* patterns: `distinct` DISTINCT patterns of m bytes each, back to back.  NULL: not a set the engine takes / out of memory. */
struct smh_hashes *smh_hash_build(const unsigned char *patterns, int m, int distinct, const char **why)
{
    const char *dummy;
    if (!why) why = &dummy;
    *why = "";
    if (m < SMH_HASH_MIN_M || m > SMH_HASH_MAX_M || distinct < 1) { *why = "pattern length outside 4..32"; return NULL; }
    if ((uint32_t)distinct > (1u << 20)) { *why = "more than 2^20 patterns"; return NULL; }
    struct smh_hashes *k = (struct smh_hashes *)calloc(1, sizeof *k);
    if (!k) { *why = "out of memory"; return NULL; }
    struct smh_hash_params P;
    memset(&P, 0, sizeof P);
    P.m = m;
    uint32_t bm = 1;
    for (int i = 0; i < m; ++i) bm = (uint32_t)(((uint64_t)bm * SMH_HASH_BASE) & 0xFFFFFFu);
    P.neg_bm = (0x1000000u - bm) & 0xFFFFFFu;
    /* filter: 10 bits per key when LDS allows, 2^15 words (128 KiB) at most, 2^8 at least */
    int wl = 8;
    while (wl < 15 && (32u << wl) < 10u * (uint32_t)distinct) ++wl;
    P.bloom_shift = (uint32_t)(24 - wl - 2);
    P.bloom_mask = ((1u << wl) - 1u) << 2;
    P.bloom_bytes = 4u << wl;
    P.bit2_shift = smh_hash_bit2_shift((uint32_t)wl);
    P.slot_dwords = ((uint32_t)m + 3u) / 4u;
    /* buckets of two slots, two tables: 4 N slots for `distinct` patterns at 82 % */
    uint32_t N = (uint32_t)((double)distinct / (4.0 * 0.82)) + 4u;
    P.slots = N;
    const size_t slot_bytes = 4u * (size_t)P.slot_dwords;
    k->table_bytes = 4u * (size_t)N * slot_bytes;
    k->table = (unsigned char *)calloc(1, k->table_bytes + 64);
    uint32_t *slot_of = (uint32_t *)calloc(4u * (size_t)N, sizeof(uint32_t)); /* [2 * bucket + slot], buckets of table 2 behind table 1's */
    if (!k->table || !slot_of) { free(slot_of); smh_hash_free(k); *why = "out of memory"; return NULL; }
    /* The filter, with two bits per window and with three (round 6).  A non-matching window passes when all of its bits are set:
     * about (bits set / bits)^k, more for the uneven words -- SAMPLED on the finished filter (65536 pseudo-random hashes), not
     * modelled.  The third bit costs the scan four vector instructions per column (0.36 -> 0.43 ms/GiB with the candidates dropped)
     * and spares stage 2 the candidates it removes: 100 000 byte patterns 3.9 % -> 3.0 % of the non-matching windows, 0.71 -> 0.61
     * ms/GiB on uniform text and 0.88 / 0.95 / 1.30 -> 0.84 / 0.89 / 1.12 on natural-language-like text (m = 8 / 12 / 20), while 30 000
     * patterns (1.7 % -> 1.1 %) lose 4 % (profiles/r06_final/notes/ab_hash_third_bit.log): whichever estimate is lower is built. */
    const int forced_k = smh_tune_int(SMH_TUNE_HASH, "bits=", 0); /* testing library only: "bits=2|3" */
    double best_ms = 0.0;
    for (uint32_t bits = 2; bits <= 3u; ++bits) {
        if (forced_k >= 2 && forced_k <= 3 && (uint32_t)forced_k != bits) continue;
        uint32_t *bloom = (uint32_t *)calloc(1, P.bloom_bytes);
        if (!bloom) { free(slot_of); smh_hash_free(k); *why = "out of memory"; return NULL; }
        for (int j = 0; j < distinct; ++j) {
            const uint32_t h = hash_roll(patterns + (size_t)j * (size_t)m, m);
            bloom[smh_hash_word_addr(h, P.bloom_shift, P.bloom_mask) >> 2] |= (1u << (h & 31u)) | (1u << smh_hash_bit2(h, P.bit2_shift)) | (bits >= 3u ? 1u << smh_hash_bit3(h) : 0u);
        }
        uint32_t pass = 0, x = 0x2545F491u;
        for (int i = 0; i < 65536; ++i) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            const uint32_t h = x & 0xFFFFFFu, w = bloom[smh_hash_word_addr(h, P.bloom_shift, P.bloom_mask) >> 2];
            pass += (w >> (h & 31u)) & (w >> smh_hash_bit2(h, P.bit2_shift)) & (bits >= 3u ? w >> smh_hash_bit3(h) : 1u) & 1u;
        }
        const double rate = (double)pass / 65536.0;
        const double ms = (bits >= 3u ? SMH_HASHES_MS_SCAN3 : SMH_HASHES_MS_SCAN) + SMH_HASHES_MS_PER_SURVIVOR * 4096.0 * rate;
        if (!k->bloom || ms < best_ms) {
            free(k->bloom);
            k->bloom = bloom;
            k->pass_rate = rate;
            P.bloom_k = bits;
            best_ms = ms;
        } else {
            free(bloom);
        }
    }
    /* cuckoo placement of the patterns (random walk); a set that does not place is retried under another seed */
    int ok = 0;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    for (uint32_t attempt = 0; attempt < 16u && !ok; ++attempt) {
        P.seed = attempt * 0x7F4A7C15u;
        memset(slot_of, 0, sizeof(uint32_t) * 4u * (size_t)N);
        ok = 1;
        for (int j = 0; j < distinct && ok; ++j) {
            uint32_t cur = (uint32_t)j + 1;
            int done = 0;
            for (uint32_t kicks = 0; kicks < 4000u && !done; ++kicks) {
                uint32_t b1, b2;
                smh_hash_slots(hash_tag(patterns + (size_t)(cur - 1) * (size_t)m, m), P.seed, N, &b1, &b2);
                const uint32_t cand[4] = {2u * b1, 2u * b1 + 1u, 2u * b2, 2u * b2 + 1u};
                for (int c = 0; c < 4 && !done; ++c)
                    if (!slot_of[cand[c]]) { slot_of[cand[c]] = cur; done = 1; }
                if (done) break;
                rng = rng * 6364136223846793005ull + 1442695040888963407ull;
                const uint32_t victim = cand[(rng >> 33) & 3u];
                const uint32_t out = slot_of[victim];
                slot_of[victim] = cur;
                cur = out;
            }
            ok = done;
        }
    }
    if (!ok) { free(slot_of); smh_hash_free(k); *why = "no cuckoo placement found"; return NULL; }
    /* slots: the pattern zero-padded to the slot; a free slot holds a string whose own buckets are other ones, so that no
     * window that is looked up here can equal it */
    for (uint32_t s = 0; s < 4u * N; ++s) {
        unsigned char f[SMH_HASH_MAX_M + 4];
        memset(f, 0, sizeof f);
        if (slot_of[s]) {
            memcpy(f, patterns + (size_t)(slot_of[s] - 1) * (size_t)m, (size_t)m);
        } else {
            for (uint32_t v = 0;; ++v) {
                memset(f, 0, sizeof f);
                memcpy(f, &v, 4);
                uint32_t b1, b2;
                smh_hash_slots(hash_tag(f, m), P.seed, N, &b1, &b2);
                if (b1 != s / 2u && b2 != s / 2u) break;
            }
        }
        /* the bucket's two slots interleaved dword by dword: slot k's dword j at dword 2 j + k of the bucket */
        unsigned char *bucket = k->table + (size_t)(s / 2u) * 2u * slot_bytes;
        for (uint32_t j = 0; j < P.slot_dwords; ++j) memcpy(bucket + 4u * (2u * j + (s & 1u)), f + 4u * j, 4);
    }
    free(slot_of);
    k->magic = SMH_MAGIC_HASHES;
    k->m = m;
    k->distinct = (uint32_t)distinct;
    k->P = P;
    /* scan 0.43 ms/GiB (twelve VALU and one LDS read per column) + two round trips per surviving column: measured (round 5) */
    k->ms_est = (P.bloom_k >= 3u ? SMH_HASHES_MS_SCAN3 : SMH_HASHES_MS_SCAN) + SMH_HASHES_MS_PER_SURVIVOR * 4096.0 * k->pass_rate;
    return k;
}

/* the engine's two tests on the host (tests): does the window pass the filter / is it a stored pattern */
int smh_hash_filter_passes(const struct smh_hashes *k, const unsigned char *window)
{
    const uint32_t h = hash_roll(window, k->m);
    const uint32_t w = k->bloom[smh_hash_word_addr(h, k->P.bloom_shift, k->P.bloom_mask) >> 2];
    return (int)((w >> (h & 31u)) & (w >> smh_hash_bit2(h, k->P.bit2_shift)) & (k->P.bloom_k >= 3u ? w >> smh_hash_bit3(h) : 1u) & 1u);
}
int smh_hash_contains(const struct smh_hashes *k, const unsigned char *window)
{
    uint32_t s1, s2;
    smh_hash_slots(hash_tag(window, k->m), k->P.seed, k->P.slots, &s1, &s2);
    const size_t sb = 4u * (size_t)k->P.slot_dwords;
    unsigned char pad[SMH_HASH_MAX_M + 4];
    memset(pad, 0, sizeof pad);
    memcpy(pad, window, (size_t)k->m);
    for (uint32_t slot = 0; slot < 4u; ++slot) {
        const unsigned char *bucket = k->table + (size_t)(slot < 2u ? s1 : s2) * 2u * sb;
        int same = 1;
        for (uint32_t j = 0; j < k->P.slot_dwords && same; ++j) same = memcmp(bucket + 4u * (2u * j + (slot & 1u)), pad + 4u * j, 4) == 0;
        if (same) return 1;
    }
    return 0;
}
