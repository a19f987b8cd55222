/*
 * csrc/hash_engine.h -- the arithmetic shared by the window-hash engine's host builder (hash_host.c) and its lane code
 * (hash_lane.h).
 *
 * The window-hash engine (round 5) is the filter engine for sets too large for the key engine (key_hash.h) on text that defeats
 * q-gram filters: 100 000 byte patterns sampled from natural-language-like text let half of all columns through a 3-byte-gram
 * filter (bench.py skewed.ascii_skewed: 2000 surviving columns per 4 KiB), because the text's common grams ARE the patterns'
 * grams.  A filter keyed on the WHOLE m-byte window has no such dependence: its false-positive rate is the load of its table
 * whatever the text.
 *   stage 1 (LDS, every column): a polynomial hash of the window rolled along the text -- h = h * B + in - out * B^m mod 2^24,
 *           two v_mad_u32_u24 -- indexes a blocked Bloom filter, two bits in one 32-bit word (ds_read_b32): 100 000 keys in 2^20
 *           bits let ~4 % of non-matching columns through.
 *   stage 2 (L2 / Infinity Cache, surviving columns, compacted per wave): the window's aligned dwords are requested from global
 *           memory (streamed microseconds ago), hashed again with the verify stage's multiply-xorshift, and BOTH of the window's
 *           buckets (two slots each) in a two-table cuckoo hash of the PATTERNS THEMSELVES (zero-padded to whole dwords) are compared
 *           with it in registers: two dependent round trips for every surviving column, true match or not -- no tag, no third trip to
 *           the pattern array (wm_lane.h smh_wm_probe_from pays one for every true match, and on such text one column in twenty
 *           is a true match).
 * Exact: a column counts when its window equals a stored pattern (what wu/wu.c:88's memcmp decides).
 */
#ifndef SMH_HASH_ENGINE_H
#define SMH_HASH_ENGINE_H

#include <stdint.h>

#define SMH_HASH_BASE 0x5BD1E9u    /* B: odd, 24 bits */
#define SMH_HASH_MAX_M 32          /* the window's dwords: nine aligned ones cover 33 bytes at any alignment */
#define SMH_HASH_MIN_M 4
#define SMH_HASH_MUL3 0xC2B2AFu    /* the third filter bit's index: bits 21..25 of the low 24 bits of h times this (one v_mul_u32_u24) */

struct smh_hash_params {
    int m;
    uint32_t neg_bm;       /* 2^24 - B^m mod 2^24: h += out * neg_bm removes the byte that leaves the window */
    uint32_t bloom_shift;  /* byte address of the window's filter word = (h >> bloom_shift) & bloom_mask */
    uint32_t bloom_mask;   /* (words - 1) << 2 */
    uint32_t bloom_bytes;  /* LDS image: 2^15 words at most */
    uint32_t slots;        /* BUCKETS per pattern table (two tables); a bucket = two slots back to back: 85 % of the slots hold a pattern,
                            * where one-slot buckets place 42 % -- half the table, and the probes are random 128-byte line fills of
                            * a table that should stay in L2 */
    uint32_t seed;         /* of the slot hashes: the builder retries with another one when the patterns do not place */
    uint32_t bloom_k;      /* filter bits per window: 2, or (round 6) 3 -- the third one at bit smh_hash_bit3(h) of the same word */
    uint32_t bit2_shift;   /* the second filter bit's index = (h >> bit2_shift) & 31: 5, or 4 in the 2^15-word filter (below) */
    uint32_t slot_dwords;  /* (m + 3) / 4: a slot is the pattern zero-padded to whole dwords, slots back to back (a table that is a third smaller
                            * than with 16 / 32-byte slots stays in L2 that much better: the probes are random 128-byte line fills) */
};

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
#define SMH_HASH_FN __device__ __forceinline__
SMH_HASH_FN uint32_t smh_hash_mad24(uint32_t a, uint32_t b, uint32_t c) { return __umul24(a, b) + c; }
#else
#define SMH_HASH_FN static inline
SMH_HASH_FN uint32_t smh_hash_mad24(uint32_t a, uint32_t b, uint32_t c) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)) + c; }
#endif

/* one byte enters the rolling hash (only its low 24 bits mean anything) */
SMH_HASH_FN uint32_t smh_hash_in(uint32_t h, uint32_t byte) { return smh_hash_mad24(h, SMH_HASH_BASE, byte); }
/* ... and the byte m places back leaves it */
SMH_HASH_FN uint32_t smh_hash_out(uint32_t h, uint32_t byte, uint32_t neg_bm) { return smh_hash_mad24(byte, neg_bm, h); }
/* the window's filter bits: bit (h & 31), bit ((h >> bit2_shift) & 31) and -- bloom_k = 3 -- bit smh_hash_bit3(h) of the word at byte address
 * (h >> shift) & mask.  The word address is the TOP bits of the 24 (2^15 words: bits 9..23), and the second index must end below it:
 * bits 5..9 for every size until late in round 6, so that in the largest filter every word's second bits fell into the half that
 * address bit 9 names -- 100 000 patterns passed 5.0 % of random windows where independent bits pass 3.8 %.  That filter now takes
 * bits 4..8 (one bit shared with the first index: 3.9 %); the smaller ones keep bits 5..9, which they do not address with. */
SMH_HASH_FN uint32_t smh_hash_bit2_shift(uint32_t words_log2) { return words_log2 >= 15u ? 4u : 5u; }
SMH_HASH_FN uint32_t smh_hash_bit2(uint32_t h, uint32_t bit2_shift) { return (h >> bit2_shift) & 31u; }
/* third index: bits 21..25 of (low 24 bits of h) x SMH_HASH_MUL3.  (First build of round 6: the product's top five bits -- within a
 * word only h's low nine bits vary, their share of the product's top bits is nearly linear in them, and so is the second index:
 * the third bit followed the second.  Five bits lower the low bits' share wraps many times: 100 000 patterns 3.5 % -> 3.0 % of
 * random windows pass, against 2.6 % for three independent bits, which nine free bits per word cannot give.) */
SMH_HASH_FN uint32_t smh_hash_bit3_raw(uint32_t h) { return smh_hash_mad24(h, SMH_HASH_MUL3, 0u) >> 21; } /* the index = its low five bits */
SMH_HASH_FN uint32_t smh_hash_bit3(uint32_t h) { return smh_hash_bit3_raw(h) & 31u; }
SMH_HASH_FN uint32_t smh_hash_word_addr(uint32_t h, uint32_t shift, uint32_t mask) { return (h >> shift) & mask; }

/* the two slots of a window whose verify-stage hash (wm_lane.h smh_wm_tag_dwords == wm_host.c smh_wm_tag) is `tag` */
SMH_HASH_FN uint32_t smh_hash_mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
SMH_HASH_FN void smh_hash_slots(uint32_t tag, uint32_t seed, uint32_t slots, uint32_t *s1, uint32_t *s2)
{
    uint32_t t1 = (tag ^ seed) * 0x9E3779B1u;
    t1 ^= t1 >> 15;
    uint32_t t2 = (tag + seed) * 0x85EBCA6Bu;
    t2 ^= t2 >> 13;
    *s1 = smh_hash_mulhi(t1 * 0x2C1B3C6Du, slots);
    *s2 = slots + smh_hash_mulhi(t2 * 0xC2B2AE35u, slots);
}

#endif
