/*
 * csrc/key_host.c -- host builder of the key engine (key_hash.h): the distinct patterns of one length as a two-table
 * cuckoo hash of their keys, laid out as the LDS image the kernels stage (key_kernels.hip).
 *
 * Replaces, for the sets it takes, what ac/ac.c:127-196 (ac_addstring) + :79-124 (ac_maketree) build and ac/ac.c:198-222
 * (search_ac) walks: the count is |{e : text[e-m+1 .. e] in set(patterns)}| (SURVEY 8a "result definition"), and a window
 * is in the set exactly when its key sits in one of its two slots.
 */
#include <stdlib.h>
#include <string.h>
#include "smh_internal.h"
#include "key_hash.h"

static uint64_t splitmix(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static int ceil_log2(uint32_t v)
{
    int b = 0;
    while ((1u << b) < v) ++b;
    return b;
}

int smh_keys_symbol_bits(int alphabet)
{
    int b = ceil_log2((uint32_t)alphabet);
    return b < 2 ? 2 : b;
}

static int cmp_u64(const void *a, const void *b)
{
    const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

static uint32_t key_f(uint64_t key, const struct smh_key_params *K)
{
    return K->wide == 1 ? smh_key_poly(key, K) : (uint32_t)key;
}
static uint32_t key_y(uint64_t key, const struct smh_key_params *K) { return K->wide == 2 ? (uint32_t)(key >> 32) : 0u; }

/* both slots of a key as indices into slot_of[]: [0, N) = table 1, [N, 2N) = table 2 */
static void key_slot_ids(uint64_t key, const struct smh_key_params *K, uint32_t *s1, uint32_t *s2)
{
    const uint32_t wsh = K->wide == 1 ? 3u : 2u;
    uint32_t o1, o2;
    smh_key_slots(key_f(key, K), key_y(key, K), K, &o1, &o2);
    *s1 = o1 >> wsh;
    *s2 = K->slots + K->pad + ((o2 - K->base2) >> wsh);
}

/* place every key in one of its two slots (random-walk cuckoo insertion); slot contents are indices into keys[] + 1,
 * 0 = free.  0 = some key could not be placed under these multipliers. */
static int cuckoo_place(const uint64_t *keys, uint32_t n, const struct smh_key_params *K, uint32_t *slot_of /* [2 * slots] */)
{
    memset(slot_of, 0, sizeof(uint32_t) * 2u * (K->slots + K->pad));
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t cur = i + 1, s1, s2;
        key_slot_ids(keys[i], K, &s1, &s2);
        if (!slot_of[s1]) { slot_of[s1] = cur; continue; }
        if (!slot_of[s2]) { slot_of[s2] = cur; continue; }
        uint32_t pos = (i & 1u) ? s2 : s1;
        int done = 0;
        for (uint32_t kicks = 0; kicks < 1000u && !done; ++kicks) {
            const uint32_t out = slot_of[pos];
            slot_of[pos] = cur;
            if (!out) { done = 1; break; }
            cur = out; /* the evicted key moves to its other slot */
            key_slot_ids(keys[cur - 1], K, &s1, &s2);
            pos = pos == s1 ? s2 : s1;
        }
        if (!done) return 0;
    }
    return 1;
}

void smh_keys_free(struct smh_keys *k)
{
    if (!k) return;
    smh_keys_dev_free(k->dev);
    free(k->image);
    free(k);
}

/* NULL when the set is not one the engine takes (m * bits > 64, more keys than the LDS budget holds, a symbol outside the
 * alphabet, no placement found); *why (optional) then says which */
struct smh_keys *smh_keys_build(const unsigned char *patterns_flat, int m, int p_size, int alphabet, uint32_t lds_budget, const char **why)
{
    const char *dummy;
    if (!why) why = &dummy;
    *why = "";
    const int bits = smh_keys_symbol_bits(alphabet);
    if (m < 1 || p_size < 1 || alphabet < 2 || alphabet > 256) { *why = "bad arguments"; return NULL; }
    if (m * bits > SMH_KEY_MAX_BITS) { *why = "m * bits per symbol > 64"; return NULL; }
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)p_size);
    if (!keys) { *why = "out of memory"; return NULL; }
    for (int j = 0; j < p_size; ++j) {
        uint64_t key = 0;
        for (int i = 0; i < m; ++i) {
            const unsigned c = patterns_flat[(size_t)j * (size_t)m + (size_t)i];
            if (c >= (unsigned)alphabet) { free(keys); *why = "pattern symbol outside the alphabet"; return NULL; }
            key = (key << bits) | c;
        }
        keys[j] = key;
    }
    qsort(keys, (size_t)p_size, sizeof(uint64_t), cmp_u64);
    uint32_t n = 0;
    for (int j = 0; j < p_size; ++j)
        if (n == 0 || keys[j] != keys[n - 1]) keys[n++] = keys[j];

    struct smh_key_params K;
    memset(&K, 0, sizeof K);
    K.m = m;
    K.bits = bits;
    const int kb = m * bits;
    K.wide = kb <= 32 ? 0 : (kb <= SMH_KEY_QUOT_BITS ? 2 : 1);
    K.pad = K.wide == 2 ? 1u << (kb - 32) : 0u;
    K.mask_lo = kb >= 32 ? 0xFFFFFFFFu : (1u << kb) - 1u;
    K.mask_hi = kb <= 32 ? 0u : (kb >= 64 ? 0xFFFFFFFFu : (1u << (kb - 32)) - 1u);
    const uint32_t W = K.wide == 1 ? 8u : 4u;
    /* slots per table: 42 % full when LDS allows, never more than 48.5 % (two-choice cuckoo places up to 50 %) */
    uint32_t N = (uint32_t)((double)n / (2.0 * 0.42)) + 1u;
    const uint32_t cap = ((lds_budget / (2u * W)) - K.pad) & ~1u;
    if (N > cap) N = cap;
    if (N < 16u) N = 16u;
    if (N < 4u * K.pad) N = 4u * K.pad; /* quotient keys: a free slot's filler must hash outside the pad slots in front of it -- possible only while the table is longer than its padding */
    if (N > cap) { free(keys); *why = "more keys than two tables in LDS hold"; return NULL; }
    if (N > 65534u) N = 65534u; /* rounded up to even below: stays under 65536 (slots << 8 must fit the 24-bit multiply, key_hash.h) */
    if ((double)n > 0.485 * 2.0 * (double)N) { free(keys); *why = "more keys than two tables in LDS hold"; return NULL; }
    N = (N + 1u) & ~1u; /* table 2 starts 8-byte aligned */
    K.slots = N;
    K.base2 = (N + K.pad) * W;
    K.bytes = (2u * (N + K.pad) * W + 15u) & ~15u;
    const uint32_t T = N + K.pad; /* slots of one table in the image */

    uint32_t *slot_of = (uint32_t *)malloc(sizeof(uint32_t) * 2u * (size_t)T);
    struct smh_keys *k = (struct smh_keys *)calloc(1, sizeof *k);
    unsigned char *image = (unsigned char *)calloc(1, K.bytes);
    if (!slot_of || !k || !image) { free(keys); free(slot_of); free(k); free(image); *why = "out of memory"; return NULL; }
    uint64_t seed = 0x5EED5EEDull;
    int placed = 0;
    for (int t = 0; t < SMH_KEY_TRIES && !placed; ++t) {
        for (int q = 0; q < 4; ++q) K.mul[q] = ((uint32_t)splitmix(&seed) & 0xFFFFFFu) | 0x800001u; /* odd, top bit set */
        K.fold[0] = ((uint32_t)splitmix(&seed) & 0xFFFFFFu) | 0x800001u; /* B of the 64-bit keys' rolled hash, and 2^24 - B^m */
        uint32_t bm = 1;
        for (int i = 0; i < m; ++i) bm = (uint32_t)(((uint64_t)bm * K.fold[0]) & 0xFFFFFFu);
        K.fold[1] = (0x1000000u - bm) & 0xFFFFFFu;
        placed = cuckoo_place(keys, n, &K, slot_of);
    }
    if (!placed) { free(keys); free(slot_of); free(k); free(image); *why = "no cuckoo placement found"; return NULL; }
    /* the image.  A free slot holds a value that can never be read as a match: a value with a bit outside the key's mask
     * when there is one; when the stored part fills the slot (m * bits == 32 or 64; a quotient key's low half), one that no
     * probe landing on this slot carries -- a key that does not hash to it (whatever its high bits, for quotient keys). */
    const int full = kb == 32 || kb == 64 || K.wide == 2;
    int filled = 1;
    for (uint32_t s = 0; s < 2u * T && filled; ++s) {
        uint64_t v;
        if (slot_of[s]) {
            v = keys[slot_of[s] - 1];
        } else if (!full) {
            v = ~0ull;
        } else {
            for (v = 0; v < (1u << 16); ++v) {
                uint32_t s1, s2;
                key_slot_ids(v, &K, &s1, &s2); /* v < 2^32: high bits 0, so s_t = the slot of y = 0; probes with this x land on s_t .. s_t + pad - 1 */
                const uint32_t mine = s < T ? s1 : s2;
                if (K.wide == 2 ? (s < mine || s >= mine + K.pad) : mine != s) break;
            }
            if (v == (1u << 16)) filled = 0; /* (cannot happen while slots > pad: one value in slots / (slots - pad) qualifies) */
        }
        if (K.wide == 1) memcpy(image + 8u * (size_t)s, &v, 8);
        else { const uint32_t v32 = (uint32_t)v; memcpy(image + 4u * (size_t)s, &v32, 4); }
    }
    free(slot_of);
    free(keys);
    if (!filled) { free(k); free(image); *why = "no filler for a free slot"; return NULL; }
    k->magic = SMH_MAGIC_KEYS;
    k->alphabet = alphabet;
    k->m = m;
    k->n_keys = n;
    k->P = K;
    k->image = image;
    /* one column = key roll + two hashes + two LDS reads + two compares whatever text and set (measured on MI355X, round 5) */
    k->ms_est = K.wide == 1 ? SMH_KEYS_MS_WIDE : (K.wide == 2 ? SMH_KEYS_MS_QUOT : SMH_KEYS_MS_NARROW);
    return k;
}

/* the membership test itself, on the host: what the kernels compute per column (tests, and the bounds-checked paths' model) */
int smh_keys_contains(const struct smh_keys *k, uint64_t key)
{
    uint32_t o1, o2;
    smh_key_slots(key_f(key, &k->P), key_y(key, &k->P), &k->P, &o1, &o2);
    const unsigned char *im = (const unsigned char *)k->image;
    if (k->P.wide == 1) {
        uint64_t a, b;
        memcpy(&a, im + o1, 8);
        memcpy(&b, im + o2, 8);
        return a == key || b == key;
    }
    uint32_t a, b;
    memcpy(&a, im + o1, 4);
    memcpy(&b, im + o2, 4);
    return a == (uint32_t)key || b == (uint32_t)key;
}
