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
    /* ---- round 6: the BUCKET image (layout 1; below) -- one LDS read per column instead of two ---- */
    uint32_t layout;      /* 0 = the two-table cuckoo image above; 1 = bucket image */
    uint32_t bk_mul;      /* A << bk_sh, 24 bits: the rolled image H = (H << bits) + symbol * bk_mul  (mod 2^32) */
    uint32_t bk_sh;       /* 32 - bk_r * bits: H's low bk_sh bits are always zero */
    uint32_t bk_r;        /* symbols H covers: min(m, floor(32 / bits)), or floor(30 / bits) when the window is longer than that */
    uint32_t bk_old;      /* 0: m <= bk_r, H is the whole window.  else = bk_r: the older m - bk_r symbols come from the H of bk_r columns ago */
    uint32_t bk_q;        /* F = (H_old << bk_q) + H: the older symbols' image lands in F's top (m - bk_r) * bits bits */
    uint32_t bk_log2;     /* primary table: 2^bk_log2 buckets of 8 bytes {S0, S1} at offset 0; bucket = F >> (32 - bk_log2) */
    uint32_t bk2_log2;    /* overflow table: 2^bk2_log2 buckets of 16 bytes (four slots) at bk2_base; bucket = (F + (F << bk2_z)) >> (32 - bk2_log2) */
    uint32_t bk2_z, bk2_base;
    uint32_t bk_sentinel; /* S0 == sentinel: the bucket holds three keys or more -- S0 is no key, S1 is one of them, the others sit in the overflow table */
    uint32_t bk_symmask;  /* ((1 << bits) - 1) * 0x01010101: text dwords are masked once, then a symbol is a byte */
    uint32_t bk_overflow; /* keys in the overflow table (statistics) */
    uint32_t bk_crowded;  /* buckets that hold a sentinel */
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

/* ---- The bucket image (round 6).
 *
 * The cuckoo image costs a column two random LDS reads (2 x 6.7 LDS-array cycles per wave-lookup, counters in DESIGN.md 9) and 12
 * vector instructions, both units full.  This image costs ONE 8-byte read and 7-8 instructions:
 *
 *   H  = sum over the window's last r symbols of symbol_i * A * 2^(sh + bits * i)  (mod 2^32), i = 0 the newest.  Rolled along
 *        the text in two instructions, (H << bits) + symbol * (A << sh): what is older than r symbols has left through the top
 *        (r * bits + sh = 32).  A is odd, so H >> sh is a BIJECTIVE image of those r symbols: equal H, equal symbols.
 *   F  = H when the window is r symbols or fewer.  Longer windows (e = m - r more symbols, e * bits <= bk_log2): the H of r
 *        columns ago holds, in its bits [sh, sh + e * bits), a bijective image E of exactly those e older symbols (the low bits of
 *        the rolled sum depend on the newest symbols only); F = (H_old << q) + H adds E into H's top e * bits bits.
 *   bucket = the top bk_log2 bits of F; each of its two slots holds the H of a key.  "slot == H" in the bucket of F says that
 *        H and the top bits of F are a key's, hence E, hence all m symbols: exact, no verify stage.
 *   Three keys or more in a bucket (1.4 % of the buckets at 8000 keys in 2^14): slot 0 holds the sentinel, slot 1 one key, the
 *        rest go to a small overflow table of four-slot buckets indexed by the top bits of F * (2^z + 1) -- an odd multiple, so the
 *        same argument makes "slot == H" there exact while its bucket bits are at least e * bits.  Only lanes that read a sentinel
 *        look there, behind a wave-uniform branch.
 *   Free slots hold a value no probe of that bucket carries: 2 when sh >= 2 (every H has its low bits clear), else (sh < 2 only
 *        when F = H) one whose bucket bits differ.  Sentinel: 1 when sh >= 2, else 0 -- and the builder keeps a key whose H is 0 out
 *        of slot 0. */
SMH_KEY_FN uint32_t smh_keyb_roll(uint32_t H, uint32_t sym, uint32_t bits, uint32_t mul) { return (H << bits) + smh_key_mul24(sym, mul); }
SMH_KEY_FN uint32_t smh_keyb_mix(uint32_t H, uint32_t Hold, const struct smh_key_params *K) { return K->bk_old ? (Hold << K->bk_q) + H : H; }
SMH_KEY_FN uint32_t smh_keyb_off1(uint32_t F, const struct smh_key_params *K) { return (F >> (29u - K->bk_log2)) & ~7u; }
SMH_KEY_FN uint32_t smh_keyb_off2(uint32_t F, const struct smh_key_params *K)
{
    const uint32_t G = (F << K->bk2_z) + F;
    return ((G >> (28u - K->bk2_log2)) & ~15u) | K->bk2_base;
}

#endif
