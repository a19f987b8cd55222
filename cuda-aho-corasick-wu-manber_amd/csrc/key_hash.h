/*
 * csrc/key_hash.h -- the arithmetic the key engine's host builder (key_host.c) and its lane code (key_lane.h) share.
 *
 * The key engine (round 5) is the exact, single-pass, text-independent engine for sets of ONE pattern length m -- what
 * the reference's API carries (smatcher.h:89-106) -- whose automaton does not fit LDS: the m-symbol window that ends at a
 * column IS a number of m * bits bits (its KEY: oldest symbol in the highest bits), the pattern set is a set of at most a
 * few ten thousand such numbers, and "does a pattern end here" is a membership test.  The set lives in LDS as a two-table
 * cuckoo hash of the keys themselves -- every key sits in slot h1(key) of table 1 or slot h2(key) of table 2 -- so a
 * column costs two independent LDS reads and two compares whatever the text and whatever the set: no filter, no verify
 * stage, no survivors.  It answers the same question as ac/ac.c:207-219 does with one state per text symbol.
 *
 * Both hashes are built from 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24 are full rate on gfx950, v_mul_lo_u32 is
 * quarter rate).  A set whose keys do not place under the first pair of multipliers is retried under others.
 */
#ifndef SMH_KEY_HASH_H
#define SMH_KEY_HASH_H

#include <stdint.h>

#define SMH_KEY_MAX_BITS 64  /* m * bits_per_symbol of the longest key */
#define SMH_KEY_TRIES 24     /* multiplier pairs tried before the builder gives up */
#define SMH_KEY_QUOT_BITS 42 /* longest quotient key: 2^10 padding slots per table */

struct smh_key_params {
    int m;
    int bits;          /* per symbol: 2 (alphabet <= 4) .. 8 */
    int wide;          /* key class: 0 = 32-bit keys (m * bits <= 32) in 4-byte slots; 1 = 64-bit keys in 8-byte slots; 2 = QUOTIENT keys
                        * of 33..42 bits in 4-byte slots: the slot holds the key's low 32 bits x, the high bits y (< 2^10) are ADDED
                        * to the slot number -- slot_t = s_t(x) + y, tables padded by 2^(bits - 32) slots -- so that "slot s_t(x) + y
                        * holds x" says the whole key is there (a stored key (x', y') sits at s_t(x') + y': x' = x makes the slot
                        * numbers differ by y' - y) */
    uint32_t mask_lo, mask_hi; /* the key's bits in the rolling code */
    uint32_t mul[4];   /* A, B, C (24 bits, odd): h1 = (f & 0xFFFFFF) * A + ((f >> 8) & 0xFFFFFF) * B, h2 = (h1 & 0xFFFFFF) * C  (mod 2^32); [3] unused */
    uint32_t fold[2];  /* 64-bit keys (wide == 1): the hash is ROLLED along the text with the window, symbol by symbol -- f = sum of
                        * sym_i * B^(m-1-i) mod 2^24 -- [0] = B (odd, 24 bits), [1] = 2^24 - B^m mod 2^24 (what removes the symbol that
                        * leaves); h1 = f, h2 = (f & 0xFFFFFF) * C.  (Until round 5's last build: f = lo + hi * C + (hi >> 8) * D and h1 mixed f
                        * again -- 8 vector instructions per column where the rolled hash takes 4; the 64-bit class is bound by them) */
    uint32_t slots;    /* per table; any number below 65536: slot_t = ((h_t & 0xFFFFFF) * slots) >> 24 (one v_mul_hi_u32_u24 with slots << 8: it takes the low 24 bits of h_t by itself) */
    uint32_t pad;      /* quotient keys: extra slots behind each table's `slots` (2^(m * bits - 32)), else 0 */
    uint32_t base2;    /* byte offset of table 2 in the image (table 1 at 0) = (slots + pad) * slot bytes */
    uint32_t bytes;    /* the image: both tables, padded to 16 */
};

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
#define SMH_KEY_FN __device__ __forceinline__
SMH_KEY_FN uint32_t smh_key_mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
/* the compiler does not form v_mul_hi_u32_u24 by itself (it takes the quarter-rate v_mul_hi_u32); b is wave-uniform */
SMH_KEY_FN uint32_t smh_key_mulhi24(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
#else
#define SMH_KEY_FN static inline
SMH_KEY_FN uint32_t smh_key_mul24(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)); }
SMH_KEY_FN uint32_t smh_key_mulhi24(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)) >> 32); }
#endif

/* 64-bit keys: one symbol enters the rolled hash / the symbol m places back leaves it (only the low 24 bits mean anything) */
SMH_KEY_FN uint32_t smh_key_poly_in(uint32_t h, uint32_t sym, uint32_t base) { return smh_key_mul24(h, base) + sym; }
SMH_KEY_FN uint32_t smh_key_poly_out(uint32_t h, uint32_t sym, uint32_t neg_bm) { return smh_key_mul24(sym, neg_bm) + h; }
/* ... and the same hash of a whole key (oldest symbol in the highest bits): what the host builder places a pattern by */
SMH_KEY_FN uint32_t smh_key_poly(uint64_t key, const struct smh_key_params *K)
{
    uint32_t h = 0;
    for (int i = K->m - 1; i >= 0; --i) h = smh_key_poly_in(h, (uint32_t)(key >> (K->bits * i)) & ((1u << K->bits) - 1u), K->fold[0]);
    return h;
}
/* The two hashes of f (a 32-bit key, or a 64-bit key folded): h1 = (f & 0xFFFFFF) * A + (f >> 8) * B -- every bit of f reaches it
 * through one of the two products -- and h2 = (h1 & 0xFFFFFF) * C, which re-spreads h1's low 24 bits: keys that share slot 1 (h1
 * within one 2^24 / slots wide range) land all over table 2.  Four VALU for both. */

/* byte offsets of a key's two slots in the image; f = the 32 bits the hashes are taken from (the key, a 64-bit key's rolled hash, a quotient
 * key's low half), y = a quotient key's high bits (else 0) */
SMH_KEY_FN void smh_key_slots(uint32_t f, uint32_t y, const struct smh_key_params *K, uint32_t *o1, uint32_t *o2)
{
    const uint32_t h1 = K->wide == 1 ? f : smh_key_mul24(f, K->mul[0]) + smh_key_mul24(f >> 8, K->mul[1]), h2 = smh_key_mul24(h1, K->mul[2]);
    const uint32_t ns = K->slots << 8, wsh = K->wide == 1 ? 3u : 2u;
    *o1 = (smh_key_mulhi24(h1, ns) + y) << wsh;
    *o2 = ((smh_key_mulhi24(h2, ns) + y) << wsh) + K->base2;
}

#endif
