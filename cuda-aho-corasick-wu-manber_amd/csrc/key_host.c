/*
 * csrc/key_host.c -- host builder of the key engine (key_hash.h): the distinct patterns of one length as a two-table
 * cuckoo hash of their keys, laid out as the LDS image the kernels stage (key_kernels.hip).
 *
 * Replaces, for the sets it takes, what ac/ac.c:127-196 (ac_addstring) + :79-124 (ac_maketree) build and ac/ac.c:198-222
 * (search_ac) walks: the count is |{e : text[e-m+1 .. e] in set(patterns)}| (SURVEY 8a "result definition"), and a window
 * is in the set exactly when its key sits in one of its two slots.
 */
#include <math.h>
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

/* ---- the bucket image (key_hash.h "The bucket image"; round 6) ---- */
struct keyb_entry { uint32_t H, F; uint64_t order; /* bucket << 32 | H */ };

/* H and F of one key (oldest symbol in the highest bits), exactly as the lane code rolls them along a text that ends with it */
static struct keyb_entry keyb_image(uint64_t key, const struct smh_key_params *K)
{
    const uint32_t smask = (1u << K->bits) - 1u;
    uint32_t H = 0, Hold = 0;
    for (int i = K->m - 1; i >= 0; --i) {
        if (K->bk_old && i == (int)K->bk_r - 1) Hold = H; /* the image after the window's first m - r symbols: what H was r columns ago */
        H = smh_keyb_roll(H, (uint32_t)(key >> (K->bits * i)) & smask, (uint32_t)K->bits, K->bk_mul);
    }
    struct keyb_entry e = { H, smh_keyb_mix(H, Hold, K), 0 };
    return e;
}

static int cmp_entry_order(const void *a, const void *b)
{
    const uint64_t x = ((const struct keyb_entry *)a)->order, y = ((const struct keyb_entry *)b)->order;
    return x < y ? -1 : x > y;
}

/* 1 = K and *image_out describe the set; 0 = this set is not one the bucket image takes (the caller falls back on the cuckoo image) */
static int keyb_build(const uint64_t *keys, uint32_t n, int m, int bits, uint32_t lds_budget, struct smh_key_params *K, unsigned char **image_out)
{
    const int r_max = 32 / bits;
    K->layout = 1;
    if (m <= r_max) {
        K->bk_r = (uint32_t)m; K->bk_old = 0; K->bk_q = 0;
    } else {
        K->bk_r = (uint32_t)(30 / bits); /* two spare low bits: free slots and the sentinel need values no H takes */
        K->bk_old = K->bk_r;
        if (K->bk_old != 15u && K->bk_old != 6u) return 0; /* the lane code keeps the delay line for 2- and 5-bit symbols (DNA, proteins) */
    }
    K->bk_sh = 32u - K->bk_r * (uint32_t)bits;
    if (K->bk_sh > 12u) return 0; /* windows under 20 bits: the multiplier would be too short to mix -- and the dense plans serve them */
    const uint32_t eb = K->bk_old ? ((uint32_t)m - K->bk_r) * (uint32_t)bits : 0u;
    if (K->bk_old) K->bk_q = 32u - K->bk_sh - eb;
    if (K->bk_old && 16u * (uint32_t)((m - 1 + 15) / 16) < K->bk_old) return 0; /* the halo primes the whole delay line */
    K->bk_sentinel = K->bk_sh >= 2u ? 1u : 0u;
    K->bk_symmask = ((1u << bits) - 1u) * 0x01010101u;
    /* primary table: 2^14 buckets (128 KiB) from a thousand keys up -- what the lanes pay for is a CROWDED bucket (three keys or
     * more: a sentinel, a queue entry, an overflow-table look), and their share falls with the cube of the load; small sets take
     * a tenth of a key per bucket (staging the image costs the launch ~0.2 us per KiB) */
    uint32_t k = n >= 1024u ? 14u : (uint32_t)ceil_log2(10u * n);
    if (k < 8u) k = 8u;
    if (k < eb) k = eb;
    if (k > 14u) k = 14u;
    k = (uint32_t)smh_tune_int(SMH_TUNE_KEY, "buckets_log2=", (int)k); /* testing library only */
    if (k > 14u || k < 4u) k = 14u;
    if (eb > k) return 0;
    K->bk_log2 = k;
    K->slots = 1u << k; /* (what the handles' info reports as key_slots / 2: buckets of two slots) */
    K->bk2_base = 8u << k;
    struct keyb_entry *ent = (struct keyb_entry *)malloc(sizeof *ent * (size_t)n);
    uint32_t *over = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n); /* indices into ent[] of the keys that go to the overflow table */
    unsigned char *fill2 = NULL;
    unsigned char *image = NULL;
    int ok = 0;
    uint64_t seed = 0xB0C4E7ull + (uint64_t)n * 2654435761ull;
    for (int t = 0; t < 48 && !ok && ent && over; ++t) {
        const uint32_t abits = 24u - K->bk_sh;
        K->bk_mul = ((((uint32_t)splitmix(&seed) & ((1u << abits) - 1u)) | (1u << (abits - 1u)) | 1u)) << K->bk_sh;
        const uint32_t shift = 32u - k;
        for (uint32_t i = 0; i < n; ++i) {
            ent[i] = keyb_image(keys[i], K);
            ent[i].order = ((uint64_t)(ent[i].F >> shift) << 32) | ent[i].H;
        }
        qsort(ent, n, sizeof *ent, cmp_entry_order);
        int dup = 0; /* two keys with the same bucket and H would mean the image is not injective: cannot happen (key_hash.h); checked all the same */
        for (uint32_t i = 1; i < n && !dup; ++i) dup = ent[i].H == ent[i - 1].H && (ent[i].F >> shift) == (ent[i - 1].F >> shift);
        if (dup) continue;
        uint32_t n_over = 0;
        for (uint32_t i = 0; i < n;) {
            uint32_t j = i;
            while (j < n && (ent[j].F >> shift) == (ent[i].F >> shift)) ++j;
            if (j - i >= 3u) for (uint32_t q = i + 1; q < j; ++q) over[n_over++] = q; /* the first stays in slot 1 */
            /* sentinel 0 (images without spare bits): H = 0 lands in bucket 0 and nowhere else -- that bucket must not overflow,
             * so that no lane ever reads a sentinel equal to its own H (key_lane.h: slot 0's compare is not masked) */
            if (j - i >= 3u && K->bk_sentinel == 0u && (ent[i].F >> shift) == 0u) dup = 1;
            i = j;
        }
        if (dup) continue;
        /* overflow table: four slots per bucket, a quarter full or less, as many bucket bits as the older symbols need */
        uint32_t k2 = (uint32_t)ceil_log2(n_over < 4u ? 4u : n_over);
        if (k2 < 4u) k2 = 4u;
        if (k2 < eb) k2 = eb;
        if (k2 > k - 1u) { if (eb > k - 1u) break; k2 = k - 1u; }
        while (k2 > 4u && k2 > eb && (8u << k) + (16u << k2) > lds_budget) --k2;
        if ((8u << k) + (16u << k2) > lds_budget) break;
        K->bk2_log2 = k2;
        K->bytes = (8u << k) + (16u << k2);
        free(image);
        image = (unsigned char *)malloc(K->bytes);
        free(fill2);
        fill2 = (unsigned char *)calloc((size_t)1 << k2, 1); /* slots taken per overflow bucket */
        if (!image || !fill2) break;
        for (uint32_t z = 1; z <= 24u && !ok; ++z) {
            K->bk2_z = z;
            memset(fill2, 0, (size_t)1 << k2);
            int fits = 1;
            for (uint32_t q = 0; q < n_over && fits; ++q) {
                const uint32_t b2 = (smh_keyb_off2(ent[over[q]].F, K) - K->bk2_base) >> 4;
                fits = ++fill2[b2] <= 4;
            }
            if (!fits) continue;
            /* free slots first, then the keys */
            uint32_t *tab = (uint32_t *)image;
            for (uint32_t b = 0; b < (1u << k); ++b) {
                const uint32_t empty = K->bk_sh >= 2u ? 2u : (((b ^ 1u) << shift) | 1u);
                tab[2u * b] = empty;
                tab[2u * b + 1u] = empty;
            }
            uint32_t *tab2 = (uint32_t *)(image + K->bk2_base);
            for (uint32_t b2 = 0; b2 < (1u << k2); ++b2) {
                uint32_t empty = 2u;
                if (K->bk_sh < 2u) /* F = H here: a value whose own overflow bucket is another one */
                    for (empty = 0; ((smh_keyb_off2(empty, K) - K->bk2_base) >> 4) == b2; ++empty) {}
                for (int q = 0; q < 4; ++q) tab2[4u * b2 + (uint32_t)q] = empty;
            }
            memset(fill2, 0, (size_t)1 << k2);
            for (uint32_t i = 0; i < n;) {
                uint32_t j = i;
                const uint32_t b = ent[i].F >> shift;
                while (j < n && (ent[j].F >> shift) == b) ++j;
                if (j - i >= 3u) {
                    tab[2u * b] = K->bk_sentinel;
                    tab[2u * b + 1u] = ent[i].H;
                    for (uint32_t q = i + 1; q < j; ++q) {
                        const uint32_t b2 = (smh_keyb_off2(ent[q].F, K) - K->bk2_base) >> 4;
                        tab2[4u * b2 + fill2[b2]++] = ent[q].H;
                    }
                } else if (j - i == 2u) {
                    /* sorted by H: a key whose H equals the sentinel (0, when sh < 2) is ent[i] -- it goes to slot 1 */
                    tab[2u * b] = ent[i + 1].H;
                    tab[2u * b + 1u] = ent[i].H;
                } else {
                    tab[2u * b + 1u] = ent[i].H;
                }
                i = j;
            }
            K->bk_overflow = n_over;
            K->bk_crowded = 0;
            for (uint32_t b = 0; b < (1u << k); ++b) K->bk_crowded += tab[2u * b] == K->bk_sentinel;
            ok = 1;
        }
    }
    free(ent);
    free(over);
    free(fill2);
    if (!ok) { free(image); return 0; }
    *image_out = image;
    return 1;
}

/* the bucket image's membership test on the host: the lane code's probe (key_lane.h smh_keyb_hit) */
static int keyb_contains(const struct smh_keys *k, uint64_t key)
{
    const struct smh_key_params *K = &k->P;
    const struct keyb_entry e = keyb_image(key, K);
    const unsigned char *im = (const unsigned char *)k->image;
    uint32_t s[2], t[4];
    memcpy(s, im + smh_keyb_off1(e.F, K), 8);
    if (s[0] != K->bk_sentinel) return s[0] == e.H || s[1] == e.H;
    memcpy(t, im + smh_keyb_off2(e.F, K), 16);
    return s[1] == e.H || t[0] == e.H || t[1] == e.H || t[2] == e.H || t[3] == e.H;
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
    /* round 6: the bucket image (one LDS read per column) for the sets it takes, the two-table cuckoo image for the others.
     * Testing build only: SMH_KEY_TUNE="layout=0|1" forces one (1: NULL when the set is not the bucket image's). */
    const int want_layout = smh_tune_int(SMH_TUNE_KEY, "layout=", -1);
    if (want_layout != 0) {
        unsigned char *bimage = NULL;
        struct smh_keys *kb_handle = (struct smh_keys *)calloc(1, sizeof *kb_handle);
        /* (the kernels keep 12 KiB of LDS for the waves' overflow queues: key_lane.h SMH_KEYB_QBYTES) */
        if (kb_handle && lds_budget > 12u * 1024u && keyb_build(keys, n, m, bits, lds_budget - 12u * 1024u, &K, &bimage)) {
            kb_handle->magic = SMH_MAGIC_KEYS;
            kb_handle->alphabet = alphabet;
            kb_handle->m = m;
            kb_handle->n_keys = n;
            kb_handle->P = K;
            kb_handle->image = bimage;
            /* measured (profiles/r06_final/notes/ab_key_bucket_image.log): 0.285 ms/GiB for the scan itself, plus what the crowded
             * buckets cost -- a column in which ANY lane of the wave reads a sentinel takes the wave through the queue code, a step
             * of eight columns with one takes it through the step's second look.  With f = the crowded buckets' share, 1500 keys
             * (f = 0.01 %) +0.015, 2500 (0.05 %) +0.045, 3500 (0.13 %) +0.065, 5000 (0.36 %) +0.15, 8000 (1.4 %) +0.25. */
            const double f = (double)K.bk_crowded / (double)(1u << K.bk_log2);
            const double p_col = 1.0 - pow(1.0 - f, 64.0), p_step = 1.0 - pow(1.0 - f, 512.0);
            kb_handle->ms_est = (K.bk_old ? SMH_KEYS_MS_BUCKET_OLD : SMH_KEYS_MS_BUCKET) + SMH_KEYS_MS_CROWDED_STEP * p_step + SMH_KEYS_MS_CROWDED_COL * p_col;
            const double cuckoo_ms = kb <= 32 ? SMH_KEYS_MS_NARROW : (kb <= SMH_KEY_QUOT_BITS ? SMH_KEYS_MS_QUOT : SMH_KEYS_MS_WIDE);
            if (want_layout == 1 || kb_handle->ms_est < cuckoo_ms) { free(keys); return kb_handle; }
            free(bimage); /* (8000 keys: 1.4 % of the buckets crowded, six columns in ten take the queue code, 0.53 against 0.41 ms/GiB: the cuckoo image it is) */
        }
        free(kb_handle);
        if (want_layout == 1) { free(keys); *why = "not a set the bucket image takes"; return NULL; }
        memset(&K, 0, sizeof K);
        K.m = m;
        K.bits = bits;
    }
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
    if (k->P.layout == 1) return keyb_contains(k, key);
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
