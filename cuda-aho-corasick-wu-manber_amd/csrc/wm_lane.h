/*
 * csrc/wm_lane.h -- what one lane of the Wu-Manber kernels does.
 *
 * Replaces the per-thread skip loops of the reference's wm_kernel1..5
 * (cuda/cuda_wm.cu:60-1058).  The quantity computed is the one search_wu
 * returns (wu/wu.c:49-107): the number of END columns e in [m-1, n) at which
 * at least one pattern equals text[e-m+1 .. e].  That count is a property of
 * each column alone -- a shift only skips columns that provably end no match --
 * so columns can be tested independently (cuda/cuda_wm.cu:69-70,136 splits the
 * columns over threads on the same grounds).
 *
 * Tuned path = the Wu-Manber stages with a device-sized block:
 *   SHIFT stage : rolling code of the last W symbols -> one bit of the LDS
 *                 filter; bit clear  <=>  SHIFT_dev[block] > 0  <=>  no pattern
 *                 ends in this block, column rejected
 *   HASH/PREFIX : surviving columns probe the HBM verify table with a hash of
 *                 the whole window and compare the pattern bytes (absent when
 *                 the block is the whole pattern and directly indexed: then a
 *                 set bit IS a match)
 * Table path = the reference tables as given: SHIFT lookup (LDS), skip loop,
 * bucket scan over {PREFIX_value, PREFIX_index}, byte compare.
 */
#ifndef SMH_WM_LANE_H
#define SMH_WM_LANE_H

#include "lane_common.h"
#include "hash_engine.h" /* smh_hash_slots: the cuckoo form of the verify table */

#define SMH_WM_HASH_MUL 0x9E3779B1u /* == SMH_HASH_MUL in smh_internal.h */
#define SMH_GRAM_MUL_DEV 0xD6E8FFu  /* == SMH_GRAM_MUL in smh_internal.h */
#define SMH_GRAM_BIG_BYTES_DEV 147392u /* == SMH_GRAM_BIG_BYTES in smh_internal.h: the table of KIND 8 (hashed byte grams, 143.9 KiB) */
#include <utility>

struct smh_wm_params {
    int m;
    int bits;            /* bits per symbol */
    uint32_t code_mask;  /* low block_symbols*bits bits of the rolling register */
    int filter_log2;     /* hashed filter: log2 of its bit count */
    int filter_k;        /* hashed filter: bits per key inside one 32-bit word (2..4) */
    int filter_le4;      /* hashed filter keyed by the block's 4 bytes as a little-endian dword (8-bit symbols) */
    int verify_log2;     /* slots = 1 << verify_log2 */
    const uint32_t *verify;      /* HBM: 16-byte buckets of four slots, slot = tag (12 bits) << 20 | pattern + 1; 0 = empty */
    const uint8_t *pat_sorted;   /* HBM: distinct patterns, each zero-padded to ((m+3)/4)*4 bytes */
    /* mixed-length sets scanned in ONE pass (smh_pset, SMH_ALGO_WM): the filter is built over the
     * patterns' last min-length symbols, and a surviving column is verified once per length class */
    int n_classes;                       /* 0 = a single-length set: verify / pat_sorted above */
    const struct smh_wm_class *classes;  /* HBM */
    const uint8_t *gram_g7;              /* HBM: pair-gram forms only, 16-bit per 7-symbol gram for the bounds-checked path: the
                                          * value G (KIND 1, up to 15 planes); grouped pairs (KIND 4): G_A | G_B << 8 */
    int gram_jb;                         /* grouped pairs: planes of the short-pattern group, 0 = none; pair form: -1 = lane 0 keeps its assumption */
    int gram_planes;                     /* pair form (KIND 1): planes J (2..15), candidate = state bit J-1 clear */
    /* grouped pairs: the verify stage's suffix index (HBM; wm_host.c smh_wm_build_gram_mixed), NULL = verify class by class */
    const uint32_t *sfx_slot;            /* [65536] records of eight dwords by the code of the column's last eight symbols: next record + 1 (in
                                          * sfx_ent), length (0 = empty), first dword of the pattern in sfx_pat, 0, its last 16 bytes END-aligned */
    const uint32_t *sfx_ent;             /* overflow records, same layout */
    const uint32_t *sfx_pat;             /* patterns END-aligned in whole dwords, zero-filled in front */
    /* round 5: the verify entries as a two-table cuckoo hash of two-slot buckets (smh_internal.h verify_ck), NULL = none: what the
     * PIPELINED probes read (smh_wm_pend_issue / _finish: staged verify with PIPE, windows from L2) -- both of a window's buckets are
     * requested together, four entries as before, and there is no "bucket full, try the next" trip */
    const uint32_t *verify_ck;
    uint32_t ck_buckets, ck_seed;
};

#define SMH_WM_MAX_CLASSES 32 /* distinct lengths of a mixed-length set scanned in one pass */
struct smh_wm_class {
    int m;
    int verify_log2;
    const uint32_t *verify;
    const uint8_t *pat_sorted;
};

/* window dword j of the m-byte window that starts at byte offset s: built from ALIGNED dword loads
 * and a funnel shift, so the stage never issues unaligned or byte-granular loads; the bytes past the
 * window in its last dword are cleared (patterns are stored zero-padded to whole dwords) */
SMH_LANE uint32_t smh_window_dword(const uint32_t *aligned, uint32_t shift_bits, int j, int m)
{
    const int rest = m - 4 * j; /* bytes of the window in this dword */
    const uint32_t lo = aligned[j];
    uint32_t v = lo >> shift_bits;
    /* the next aligned dword is touched only when the window really extends into it, so the stage
     * never reads past the aligned dword that holds the window's last byte */
    if (shift_bits && rest > 4 - (int)(shift_bits >> 3)) v |= aligned[j + 1] << (32u - shift_bits);
    if (rest < 4) v &= (1u << (8 * rest)) - 1u;
    return v;
}

/* hash of a pattern / window given as zero-padded little-endian dwords; mirrored by wm_host.c */
SMH_LANE uint32_t smh_wm_mix(uint32_t h, uint32_t v)
{
    h = (h ^ v) * 0x9E3779B1u;
    return h ^ (h >> 15);
}

/* second half of the device HASH/PREFIX stage: the bucket walk for a window whose hash `tag` is known.  The pattern
 * bytes are compared with the text in HBM (reached by true matches and one false tag in ~1000 probes only). */
SMH_LANE uint32_t smh_wm_first_bucket(uint32_t tag, const smh_wm_params &P) { return (tag * SMH_WM_HASH_MUL) >> (32 - (P.verify_log2 - 2)); }

/* have_first: `first` holds the contents of the window's first bucket, loaded earlier (software pipelining, see
 * smh_wm_pend_issue) */
SMH_LANE uint32_t smh_wm_probe_from(const uint8_t *text, uint64_t e, uint32_t tag, const smh_wm_params &P, bool have_first,
                                    smh_u32x4 first)
{
    const uint64_t s0 = e + 1 - (uint64_t)P.m;
    const uint32_t *aligned = reinterpret_cast<const uint32_t *>(text + (s0 & ~(uint64_t)3));
    const uint32_t shift_bits = (uint32_t)(s0 & 3u) * 8u;
    const int nd = (P.m + 3) >> 2;
    uint32_t b = smh_wm_first_bucket(tag, P); /* bucket of four slots */
    const uint32_t bmask = (1u << (P.verify_log2 - 2)) - 1u;
    for (bool preloaded = have_first;; preloaded = false) {
        smh_u32x4 q4 = first;
        if (!preloaded) q4 = smh_load16(reinterpret_cast<const uint8_t *>(P.verify) + 16u * (uint64_t)b);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t slot = q4.v[k]; /* 12 tag bits | pattern + 1 (20 bits); 0 = empty */
            if (slot != 0 && (slot >> 20) == (tag & 0xFFFu)) {
                const uint32_t *q = reinterpret_cast<const uint32_t *>(P.pat_sorted) + (uint64_t)((slot & 0xFFFFFu) - 1) * (uint32_t)nd;
                uint32_t diff = 0;
                for (int j = 0; j < nd; ++j) diff |= q[j] ^ smh_window_dword(aligned, shift_bits, j, P.m);
                if (diff == 0) return 1;
            }
        }
        if (q4.v[3] == 0) return 0; /* slots fill in order: a bucket with a free slot ends the search */
        b = (b + 1) & bmask;
    }
}

SMH_LANE uint32_t smh_wm_probe(const uint8_t *text, uint64_t e, uint32_t tag, const smh_wm_params &P)
{
    return smh_wm_probe_from(text, e, tag, P, false, smh_u32x4{{0, 0, 0, 0}});
}

/* the cuckoo form (P.verify_ck): the window's four candidate entries -- two buckets of two -- as one smh_u32x4 */
SMH_LANE smh_u32x4 smh_wm_ck_load(uint32_t tag, const smh_wm_params &P)
{
    uint32_t b1, b2;
    smh_hash_slots(tag, P.ck_seed, P.ck_buckets, &b1, &b2);
    smh_u32x4 q;
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    typedef uint32_t v2a __attribute__((ext_vector_type(2), aligned(8)));
    const v2a a = *reinterpret_cast<const v2a *>(P.verify_ck + 2u * (size_t)b1), b = *reinterpret_cast<const v2a *>(P.verify_ck + 2u * (size_t)b2);
    q.v[0] = a.x; q.v[1] = a.y; q.v[2] = b.x; q.v[3] = b.y;
#else
    q.v[0] = P.verify_ck[2u * (size_t)b1]; q.v[1] = P.verify_ck[2u * (size_t)b1 + 1u];
    q.v[2] = P.verify_ck[2u * (size_t)b2]; q.v[3] = P.verify_ck[2u * (size_t)b2 + 1u];
#endif
    return q;
}
/* ... and the decision over them: an entry whose 12 tag bits agree is compared with the text (as smh_wm_probe_from) */
SMH_LANE uint32_t smh_wm_ck_decide(const uint8_t *text, uint64_t e, uint32_t tag, const smh_wm_params &P, const smh_u32x4 &q4)
{
    const uint64_t s0 = e + 1 - (uint64_t)P.m;
    const uint32_t *aligned = reinterpret_cast<const uint32_t *>(text + (s0 & ~(uint64_t)3));
    const uint32_t shift_bits = (uint32_t)(s0 & 3u) * 8u;
    const int nd = (P.m + 3) >> 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t slot = q4.v[k];
        if (slot != 0 && (slot >> 20) == (tag & 0xFFFu)) {
            const uint32_t *q = reinterpret_cast<const uint32_t *>(P.pat_sorted) + (uint64_t)((slot & 0xFFFFFu) - 1) * (uint32_t)nd;
            uint32_t diff = 0;
            for (int j = 0; j < nd; ++j) diff |= q[j] ^ smh_window_dword(aligned, shift_bits, j, P.m);
            if (diff == 0) return 1;
        }
    }
    return 0;
}

/* device HASH/PREFIX stage: is text[e-m+1 .. e] one of the patterns?  Three dependent memory
 * phases (window dwords, one table slot, pattern dwords) instead of byte loops.  Kept small on
 * purpose: it is inlined into the scan kernel and must not raise its register pressure. */
SMH_LANE uint32_t smh_wm_verify(const uint8_t *text, uint64_t e, const smh_wm_params &P)
{
    const uint64_t s0 = e + 1 - (uint64_t)P.m;
    const uint32_t *aligned = reinterpret_cast<const uint32_t *>(text + (s0 & ~(uint64_t)3));
    const uint32_t shift_bits = (uint32_t)(s0 & 3u) * 8u;
    const int nd = (P.m + 3) >> 2;
    uint32_t tag = 0x811C9DC5u;
    for (int j = 0; j < nd; ++j) tag = smh_wm_mix(tag, smh_window_dword(aligned, shift_bits, j, P.m));
    return smh_wm_probe(text, e, tag, P);
}

/* The same hash from a STAGED copy of the text (LDS on the GPU): `rd(off)` returns the aligned dword at byte offset
 * `off` of a buffer that holds the bytes around the window, `s0` = offset of the window's first byte in it.  All
 * MAXD + 1 aligned dwords a window of up to 4 * MAXD bytes can touch are requested up front (independent LDS reads,
 * one round trip instead of one per dword; the buffer is padded for the ones past the window) and mixed in as far
 * as the window reaches. */
template <int MAXD>
SMH_LANE uint32_t smh_wm_tag_dwords(const uint32_t (&d)[MAXD + 1], uint32_t sh, int m);

template <int MAXD, typename RD>
SMH_LANE uint32_t smh_wm_tag_staged(RD rd, uint32_t s0, int m)
{
    const uint32_t a0 = s0 & ~3u, sh = (s0 & 3u) * 8u;
    uint32_t d[MAXD + 1];
#pragma unroll
    for (int j = 0; j <= MAXD; ++j) d[j] = rd(a0 + 4u * (uint32_t)j);
    return smh_wm_tag_dwords<MAXD>(d, sh, m);
}

/* the hash of the m-byte window that starts `sh` bits into d[0] (d[j] = consecutive aligned dwords) */
template <int MAXD>
SMH_LANE uint32_t smh_wm_tag_dwords(const uint32_t (&d)[MAXD + 1], uint32_t sh, int m)
{
    const int nd = (m + 3) >> 2;
    uint32_t tag = 0x811C9DC5u;
#pragma unroll
    for (int j = 0; j < MAXD; ++j) {
        if (j < nd) {
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
            uint32_t v = __builtin_amdgcn_alignbit(d[j + 1], d[j], sh);
#else
            uint32_t v = (uint32_t)((((uint64_t)d[j + 1] << 32) | d[j]) >> sh);
#endif
            const int rest = m - 4 * j;
            if (rest < 4) v &= (1u << (8 * rest)) - 1u;
            tag = smh_wm_mix(tag, v);
        }
    }
    return tag;
}

/* one length class of a mixed-length set: is text[e-m_c+1 .. e] one of the class's patterns? */
SMH_LANE uint32_t smh_wm_verify_class(const uint8_t *text, uint64_t e, const smh_wm_params &P, int c)
{
    const smh_wm_class k = P.classes[c];
    if (e + 1 < (uint64_t)k.m) return 0; /* no full window of this length ends here */
    smh_wm_params Pc = P;
    Pc.m = k.m;
    Pc.verify_log2 = k.verify_log2;
    Pc.verify = k.verify;
    Pc.verify_ck = nullptr; /* (the cuckoo form belongs to the suffix handle's own pattern set) */
    Pc.pat_sorted = k.pat_sorted;
    return smh_wm_verify(text, e, Pc);
}
/* number of length classes with a pattern ending at e (== the column's count in the decomposition) */
SMH_LANE uint32_t smh_wm_verify_any(const uint8_t *text, uint64_t e, const smh_wm_params &P)
{
    if (P.n_classes == 0) return smh_wm_verify(text, e, P);
    uint32_t cnt = 0;
    for (int c = 0; c < P.n_classes; ++c) cnt += smh_wm_verify_class(text, e, P, c);
    return cnt;
}

/* The number of DISTINCT patterns of a mixed-length set that end at column e, by the suffix index: the 32 bytes that end
 * at e come in as nine aligned dwords requested together, the
 * code of the last eight symbols picks the slot, and every entry of its chain is compared END-aligned, eight dwords at a
 * time (a pattern longer than 32 bytes that agrees so far is finished byte by byte).  Two dependent trips to the index
 * whatever the number of length classes.  ALL lanes of the wave call it (the chain loop and the positions append are
 * wave-wide); `valid` = this lane holds a column; e >= 31 where valid. */
SMH_LANE void smh_sfx_window(const uint32_t (&d)[9], uint32_t sh, uint32_t (&v)[8])
{
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
        v[k] = __builtin_amdgcn_alignbit(d[k + 1], d[k], sh);
#else
        v[k] = (uint32_t)((((uint64_t)d[k + 1] << 32) | d[k]) >> sh);
#endif
    }
}
/* v[k] = bytes e-31+4k .. e-28+4k of the text */
SMH_LANE uint32_t smh_wm_sfx_decide(const uint32_t (&v)[8], const uint8_t *text, uint64_t e, bool valid, const smh_wm_params &P, const smh_pos_out *po)
{
    /* bytes e-7 .. e are v[6], v[7]; oldest symbol in the highest bits, as wm_host.c packs the patterns' */
    uint32_t code = 0;
#pragma unroll
    for (int k = 6; k < 8; ++k)
#pragma unroll
        for (int b = 0; b < 4; ++b) code = (code << 2) | ((v[k] >> (8 * b)) & 3u);
    const uint32_t *rec = valid ? P.sfx_slot + 8u * (size_t)code : nullptr;
    uint32_t cnt = 0;
    while (SMH_WAVE_ANY(rec != nullptr)) {
        uint32_t hit = 0;
        if (rec) {
            const smh_u32x4 h = smh_load16(reinterpret_cast<const uint8_t *>(rec)), q = smh_load16(reinterpret_cast<const uint8_t *>(rec) + 16u);
            const uint32_t next = h.v[0], L = h.v[1], nd = (L + 3u) >> 2;
            const uint32_t front = 0xFFFFFFFFu << (8u * ((4u * nd - L) & 3u)); /* the bytes of the pattern's first dword that belong to it: the high ones */
            uint32_t diff = L == 0u || e + 1u < (uint64_t)L ? 1u : 0u; /* an empty slot; no whole window of this length ends here */
#pragma unroll
            for (int j = 0; j < 4; ++j) /* the last 16 bytes, from the record itself */
                if ((uint32_t)j < nd) {
                    uint32_t t = v[7 - j];
                    if ((uint32_t)j == nd - 1u) t &= front;
                    diff |= t ^ q.v[3 - j];
                }
            if (!diff && nd > 4u) { /* a longer pattern whose end agrees: the rest of it */
                const uint32_t *pp = P.sfx_pat + h.v[2];
#pragma unroll
                for (int j = 4; j < 8; ++j)
                    if ((uint32_t)j < nd) {
                        uint32_t t = v[7 - j];
                        if ((uint32_t)j == nd - 1u) t &= front;
                        diff |= t ^ pp[nd - 1u - (uint32_t)j];
                    }
                if (!diff && nd > 8u) {
                    const uint8_t *pb = reinterpret_cast<const uint8_t *>(pp);
                    for (uint32_t i = 32; i < L && !diff; ++i) diff = (uint32_t)(text[e - i] ^ pb[4u * nd - 1u - i]);
                }
            }
            hit = diff == 0u ? 1u : 0u;
            rec = next ? P.sfx_ent + 8u * (size_t)(next - 1u) : nullptr;
        }
        cnt += hit;
        if (po) smh_append_bits(hit, e, *po);
    }
    return cnt;
}
SMH_LANE uint32_t smh_wm_verify_sfx(const uint8_t *text, uint64_t e, bool valid, const smh_wm_params &P, const smh_pos_out *po)
{
    const uint64_t s0 = valid ? e - 31u : 0u;
    const uint32_t *al = reinterpret_cast<const uint32_t *>(text + (s0 & ~(uint64_t)3));
    const uint32_t sh = (uint32_t)(s0 & 3u) * 8u;
    uint32_t d[9], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = al[j];
    d[8] = al[sh ? 8 : 7]; /* an aligned window ends in d[7]: nothing past it is touched */
    smh_sfx_window(d, sh, v);
    return smh_wm_sfx_decide(v, text, e, valid, P, po);
}
/* block hash of the hashed filter: two 24-bit multiplies (v_mul_u32_u24 / v_mad_u32_u24 are
 * full-rate, v_mul_lo_u32 is not).  Keep in sync with smh_wm_block_hash in wm_host.c. */
SMH_LANE uint32_t smh_wm_block_hash(uint32_t key)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    return __umul24(key, 0x9E3779u) + __umul24(key >> 8, 0x85EBCBu);
#else
    return (uint32_t)((uint64_t)(key & 0xFFFFFFu) * 0x9E3779u) + (uint32_t)((uint64_t)((key >> 8) & 0xFFFFFFu) * 0x85EBCBu);
#endif
}

/* bit `pos & 31` of word (v_bfe_u32 takes the low five bits of its offset operand by itself) */
SMH_LANE uint32_t smh_bit_at(uint32_t word, uint32_t pos)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    return __builtin_amdgcn_ubfe(word, pos, 1u);
#else
    return (word >> (pos & 31u)) & 1u;
#endif
}

SMH_LANE uint32_t smh_bswap32(uint32_t v) { return __builtin_bswap32(v); }

/* hashed SHIFT stage for one key: a blocked Bloom filter with 64-bit blocks.  The block comes from
 * the top bits of the block hash (one ds_read_b64), the K bit positions from its low bits: the first
 * two index the block's low dword, the others its high dword (5 bits each; the fourth overlaps the
 * third by two bits).  32-bit blocks let 2.6 % of random keys through at 100 000 keys in 2^20 bits
 * (the load per block varies too much), 64-bit blocks 1.8 %, for the same VALU work.  K is a
 * compile-time constant on the tuned path (0 = read it from P).  Mirrored by wm_host.c. */
struct smh_u32x2 { uint32_t lo, hi; };
SMH_LANE smh_u32x2 smh_filter_block(const uint32_t *filter, uint32_t block)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    const uint2 v = *reinterpret_cast<const uint2 *>(filter + 2u * block);
    return smh_u32x2{v.x, v.y};
#else
    return smh_u32x2{filter[2u * block], filter[2u * block + 1u]};
#endif
}

template <int K>
SMH_LANE uint32_t smh_wm_filter_key(uint32_t key, const uint32_t *filter, const smh_wm_params &P)
{
    const uint32_t h = smh_wm_block_hash(key);
    const smh_u32x2 w = smh_filter_block(filter, h >> (38 - P.filter_log2));
    const int k = K ? K : P.filter_k;
    uint32_t hit = smh_bit_at(w.lo, h) & smh_bit_at(w.lo, h >> 5);
    if (k >= 3) hit &= smh_bit_at(w.hi, h >> 10);
    if (k >= 4) hit &= smh_bit_at(w.hi, h >> 13);
    return hit;
}

/* The byte-block form (filter_le4: key = the column's last four bytes as a little-endian dword, 2^20 filter bits) has its
 * own hash and bit layout, chosen for the instruction count of the scan (it was VALU-bound at 16.5 ops per column):
 *   h = key[23:0] * 0x9E3779 + key[31:24] * 0x85EBCB      block = bits 3..16 of h: (h & 0x1FFF8) IS its LDS byte address
 *   g = key[23:0] * 0xC2B2AF                              bit positions = the low five bits of g's BYTES 1, 2 (low dword
 *                                                         of the block), 3 and 0 (high dword; K >= 3, K >= 4) and of h's
 *                                                         byte 3 (low dword; K == 5)
 * so a column costs: v_alignbyte (key), v_mul_u32_u24 with a byte-3 select + v_mad_u32_u24 (h), v_mul_u32_u24 (g),
 * v_and (address), ds_read_b64, K shifts whose amount is a byte select of g (SDWA), one three-input AND, one
 * v_alignbit that shifts the answer into the survivor mask: 10 VALU at K = 3, 12 at K = 4.  Mirrored by wm_host.c. */
#define SMH_BLK_MUL_A 0x9E3779u
#define SMH_BLK_MUL_B 0x85EBCBu
#define SMH_BLK_MUL_G 0xC2B2AFu
template <int K>
SMH_LANE uint32_t smh_wm_filter_key_v2(uint32_t key, const uint32_t *filter, const smh_wm_params &P)
{
    const uint32_t lo24 = key & 0xFFFFFFu;
    const uint32_t h = (uint32_t)((uint64_t)lo24 * SMH_BLK_MUL_A) + (key >> 24) * SMH_BLK_MUL_B;
    const uint32_t g = (uint32_t)((uint64_t)lo24 * SMH_BLK_MUL_G);
    const smh_u32x2 w = smh_filter_block(filter, (h >> 3) & 0x3FFFu);
    const int k = K ? K : P.filter_k;
    uint32_t hit = smh_bit_at(w.lo, g >> 8) & smh_bit_at(w.lo, g >> 16);
    if (k >= 3) hit &= smh_bit_at(w.hi, g >> 24);
    if (k >= 4) hit &= smh_bit_at(w.hi, g);
    if (k >= 5) hit &= smh_bit_at(w.lo, h >> 24);
    return hit;
}
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
/* the same test as the instruction sequence above, in two halves so that the caller can keep several blocks' reads in
 * flight: LDS byte address of the key's block (and g), then the test (bit 0 of the result is the answer) */
SMH_LANE uint32_t smh_wm_v2_addr(uint32_t key, uint32_t &g, uint32_t &h)
{
    uint32_t t;
    const uint32_t mulb = SMH_BLK_MUL_B;
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(t) : "v"(key), "v"(mulb));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(h) : "v"(key), "s"(SMH_BLK_MUL_A), "v"(t));
    g = __umul24(key, SMH_BLK_MUL_G);
    return h & 0x1FFF8u;
}
template <int K>
SMH_LANE uint32_t smh_wm_v2_test(uint32_t g, uint32_t h, uint32_t lo, uint32_t hi)
{
    uint32_t a, b, c;
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(a) : "v"(g), "v"(lo));
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(b) : "v"(g), "v"(lo));
    uint32_t r = a & b;
    if (K >= 3) {
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(c) : "v"(g), "v"(hi));
        r &= c;
    }
    if (K >= 4) {
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(c) : "v"(g), "v"(hi));
        r &= c;
    }
    if (K >= 5) {
        asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(c) : "v"(h), "v"(lo));
        r &= c;
    }
    return r;
}
#endif

/* SHIFT stage for one column: returns 1 when the block's filter bit(s) are set */
template <bool HASHED>
SMH_LANE uint32_t smh_wm_filter(uint32_t code, const uint32_t *filter, const smh_wm_params &P)
{
    const uint32_t key = code & P.code_mask;
    if (HASHED) {
        if (P.filter_le4) return smh_wm_filter_key_v2<0>(smh_bswap32(key), filter, P); /* rolling code = big-endian */
        return smh_wm_filter_key<0>(key, filter, P);
    } else {
        return (filter[key >> 5] >> (key & 31u)) & 1u;
    }
}

/* ---- per-wave survivor queue: columns that passed the SHIFT stage wait here for the HASH/PREFIX
 * stage, which then runs with all 64 lanes busy (wavefront-level compaction: ballot + prefix
 * count).  Same scheme as the AC candidate queue (ac_lane.h). */
/* Two windows at once: the stage is three DEPENDENT memory round trips (window dwords, table slot,
 * pattern dwords) and a wave has nothing else to do while it waits, so a second independent chain
 * per lane hides about half of that latency.  Same result as two smh_wm_verify calls. */
SMH_LANE uint32_t smh_wm_probe2(const uint8_t *text, uint64_t e0, uint64_t e1, uint32_t tag0, uint32_t tag1,
                                const smh_wm_params &P, uint32_t &r1)
{
    const uint64_t b0 = e0 + 1 - (uint64_t)P.m, b1 = e1 + 1 - (uint64_t)P.m;
    const uint32_t *al0 = reinterpret_cast<const uint32_t *>(text + (b0 & ~(uint64_t)3));
    const uint32_t *al1 = reinterpret_cast<const uint32_t *>(text + (b1 & ~(uint64_t)3));
    const uint32_t sh0 = (uint32_t)(b0 & 3u) * 8u, sh1 = (uint32_t)(b1 & 3u) * 8u;
    const int nd = (P.m + 3) >> 2;
    const uint32_t bshift = 32u - (uint32_t)(P.verify_log2 - 2), bmask = (1u << (P.verify_log2 - 2)) - 1u;
    uint32_t b0q = (tag0 * SMH_WM_HASH_MUL) >> bshift, b1q = (tag1 * SMH_WM_HASH_MUL) >> bshift;
    uint32_t r0 = 0;
    bool a0 = true, a1 = true;
    r1 = 0;
    const uint8_t *vt = reinterpret_cast<const uint8_t *>(P.verify);
    for (;;) {
        /* both chains' buckets in flight together (one 16-byte load each) */
        const smh_u32x4 q0 = smh_load16(vt + 16u * (uint64_t)b0q), q1 = smh_load16(vt + 16u * (uint64_t)b1q);
        if (a0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t slot = q0.v[k];
                if (slot != 0 && (slot >> 20) == (tag0 & 0xFFFu)) {
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(P.pat_sorted) + (uint64_t)((slot & 0xFFFFFu) - 1) * (uint32_t)nd;
                    uint32_t diff = 0;
                    for (int j = 0; j < nd; ++j) diff |= q[j] ^ smh_window_dword(al0, sh0, j, P.m);
                    if (diff == 0) r0 = 1;
                }
            }
            if (r0 || q0.v[3] == 0) a0 = false;
            b0q = (b0q + 1) & bmask;
        }
        if (a1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t slot = q1.v[k];
                if (slot != 0 && (slot >> 20) == (tag1 & 0xFFFu)) {
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(P.pat_sorted) + (uint64_t)((slot & 0xFFFFFu) - 1) * (uint32_t)nd;
                    uint32_t diff = 0;
                    for (int j = 0; j < nd; ++j) diff |= q[j] ^ smh_window_dword(al1, sh1, j, P.m);
                    if (diff == 0) r1 = 1;
                }
            }
            if (r1 || q1.v[3] == 0) a1 = false;
            b1q = (b1q + 1) & bmask;
        }
        if (!SMH_WAVE_ANY(a0 || a1)) break;
    }
    return r0;
}

SMH_LANE uint32_t smh_wm_verify2(const uint8_t *text, uint64_t e0, uint64_t e1, const smh_wm_params &P, uint32_t &r1)
{
    const uint64_t b0 = e0 + 1 - (uint64_t)P.m, b1 = e1 + 1 - (uint64_t)P.m;
    const uint32_t *al0 = reinterpret_cast<const uint32_t *>(text + (b0 & ~(uint64_t)3));
    const uint32_t *al1 = reinterpret_cast<const uint32_t *>(text + (b1 & ~(uint64_t)3));
    const uint32_t sh0 = (uint32_t)(b0 & 3u) * 8u, sh1 = (uint32_t)(b1 & 3u) * 8u;
    const int nd = (P.m + 3) >> 2;
    uint32_t tag0 = 0x811C9DC5u, tag1 = 0x811C9DC5u;
    for (int j = 0; j < nd; ++j) {
        const uint32_t v0 = smh_window_dword(al0, sh0, j, P.m), v1 = smh_window_dword(al1, sh1, j, P.m);
        tag0 = smh_wm_mix(tag0, v0);
        tag1 = smh_wm_mix(tag1, v1);
    }
    return smh_wm_probe2(text, e0, e1, tag0, tag1, P, r1);
}

/* The STG template value of the gram kernels says how a surviving column gets its window hash:
 *   0      no staging: windows are re-read from HBM by the drain
 *   1, 2   staged verify: the chunk is copied to LDS (16 / 32 bytes of halo), smh_wm_stage_*
 *   5, 6   in-register verify (round 3, pair form): the lane selects the window's dwords out of its OWN text registers
 *          and the previous lane's last 16 / 32 bytes (DPP) with a barrel of conditional moves -- no copy, no lock, no
 *          queue, no LDS round trip; smh_wm_regv_columns */
/*   3, 4   windows from L2 (round 4, byte-gram forms; m - 1 <= 16 / 32): no copy of the chunk at all -- the surviving columns
 *          are compacted into the wave's queue as before, every lane takes one and REQUESTS its window's aligned dwords from
 *          global memory (the chunk was streamed a few microseconds ago: L2 / Infinity Cache), and the requests ride through
 *          the next chunk's scan; then the hash, the bucket request, another chunk's scan, the decision -- a two-stage
 *          software pipeline with no LDS round trip a wave has to wait for; smh_wm_l2_columns */
constexpr bool smh_stg_regv(int STG) { return STG >= 5; }
constexpr bool smh_stg_staged(int STG) { return STG == 1 || STG == 2; }
constexpr bool smh_stg_l2(int STG) { return STG == 3 || STG == 4; }
constexpr int smh_stg_hp(int STG) { return STG >= 5 ? STG - 4 : (STG == 1 || STG == 2 ? STG : 1); } /* 16-byte pieces of text kept from in front of the chunk */
constexpr int smh_stg_l2_maxd(int STG) { return STG == 3 ? 5 : 9; } /* == SMH_STAGE_MAXD: dwords of the longest window */

#define SMH_WM_QCAP 128u /* END columns per wave, 1 KiB of LDS behind the filter */
struct smh_wm_queue {
    uint64_t *slots; /* SMH_WM_QCAP entries, private to this wave (LDS on the GPU) */
    uint32_t count;  /* wave-uniform */
    uint32_t matches;
    const smh_pos_out *po; /* positions mode: verified columns are appended here; else NULL */
    /* staged verify (below): LDS byte offsets of the lock words and the staging buffers, number of buffers, and the
     * number of surviving columns a wave-chunk must have for its windows to be hashed from the staged copy */
    uint32_t st_locks, st_bufs, st_nbufs, st_min;
    /* software-pipelined probe (gram kernels): up to 64 hashed columns of the chunk staged last -- one per lane --
     * wait here; their buckets are requested before the next chunk is scanned and looked at after it */
    uint32_t pend_n;      /* wave-uniform: lanes below it hold a column */
    uint32_t pend_loaded; /* wave-uniform: pend_q holds the first bucket */
    uint32_t pend_tag;
    uint64_t pend_e;
    smh_u32x4 pend_q;
    uint32_t pend_mine;   /* in-register verify: per lane, this lane holds a pending column (there is no compaction) */
    uint32_t events;      /* per lane: surviving columns this lane sent to the verify stage (smh_stats.h) */
    /* windows from L2 (STG 3 / 4): the first pipeline stage -- up to 64 columns, one per lane, whose window dwords are in flight */
    uint32_t pa_n;        /* wave-uniform: lanes below it hold a column */
    uint32_t pa_sh;       /* bit offset of the window in pa_w[0] */
    uint64_t pa_e;
    uint64_t pa_limit;    /* wave-uniform: the text's length (window requests near its end take the narrow form) */
    uint32_t pa_w[10];
};

/* ---- staged verify: the window hash of a surviving column is computed from an ON-CHIP copy of the wave-chunk.
 * Re-reading the window from HBM costs one 128-byte line per surviving column -- with 100 000 byte patterns 0.75 %
 * of the columns survive any filter that fits LDS, which was 2.2x the algorithmic HBM traffic and three dependent
 * memory round trips per drain.  A wave whose chunk has st_min or more surviving columns copies its 4 KiB chunk
 * (plus 16*STG bytes in front of it) from its registers into a staging buffer in LDS, every lane takes one
 * surviving column from the queue, reads the window's dwords from the buffer at a per-lane address -- the dynamic
 * indexing registers do not offer -- and hashes them; only the bucket probe (L2-resident table) and, for a
 * matching tag, the final compare leave the CU.  Chunks with FEWER survivors leave them in the queue, which is
 * drained from HBM 64 columns at a time as before (a handful of survivors does not pay for the copy).  The table
 * fills LDS, so the workgroup's 16 waves share a few buffers through try-locks (a buffer is held for two LDS round
 * trips; whoever holds one never waits for anything else). */
#define SMH_STAGE_BUF(STG) (16u * (STG) + 4096u + 48u) /* halo, chunk, pad for the dwords read past the last window */
/* (Round 4 tried the chunk TRANSPOSED in the buffer -- piece q of lane l at q * 1024 + l * 16, so that the copy's four
 * ds_write_b128 spread over all banks instead of every other lane starting in the same one: the address arithmetic it costs
 * the window reads outweighed it, 100 000 byte patterns 1-6 % slower, DNA sets unchanged: gpurun_out/r04_i/ab_stage.log.) */
#define SMH_STAGE_MAXD(STG) ((STG) == 1 ? 5 : 9)         /* dwords of the longest window: m <= 17 / m <= 33 */
#define SMH_STAGE_MIN_DEFAULT 8u /* gpurun_out/r02_u: 1..16 within 3 % on the dense sets, 16+ better on sparse ones, 32+ loses 10-30 % */

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE void smh_wm_drain(smh_wm_queue &Q, const uint8_t *text, const smh_wm_params &P)
{
    /* at most SMH_WM_QCAP = 128 entries: lane l takes entries l and l + 64, both chains in flight
     * together; a lane with one or no entry re-verifies a column it knows to be valid and drops the
     * answer */
    if (Q.count == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    if (P.n_classes) {
        /* mixed-length set: every surviving column is looked up in the suffix index, or tried against each length class */
        for (uint32_t base = 0; base < Q.count; base += 64u) {
            const bool valid = base + lane < Q.count;
            const uint64_t e = valid ? Q.slots[base + lane] : Q.slots[0];
            if (P.sfx_slot && !SMH_WAVE_ANY(valid && e < 31u)) {
                Q.matches += smh_wm_verify_sfx(text, e, valid, P, Q.po);
                continue;
            }
            for (int c = 0; c < P.n_classes; ++c) {
                const uint32_t hit = valid ? smh_wm_verify_class(text, e, P, c) : 0u;
                Q.matches += hit;
                if (Q.po) smh_append_bits(hit, e, *Q.po);
            }
        }
        Q.count = 0;
        return;
    }
    const bool h0 = lane < Q.count, h1 = lane + 64u < Q.count;
    const uint64_t e0 = h0 ? Q.slots[lane] : Q.slots[0];
    const uint64_t e1 = h1 ? Q.slots[lane + 64u] : e0;
    uint32_t r1;
    const uint32_t r0 = smh_wm_verify2(text, e0, e1, P, r1);
    Q.matches += (h0 ? r0 : 0u) + (h1 ? r1 : 0u);
    if (Q.po) { /* positions mode: the verified END columns */
        smh_append_bits(h0 ? r0 : 0u, e0, *Q.po);
        smh_append_bits(h1 ? r1 : 0u, e1, *Q.po);
    }
    Q.count = 0;
}
/* append without a capacity check: the caller drains first whenever fewer than 64 slots are free */
SMH_LANE void smh_wm_emit(smh_wm_queue &Q, const uint8_t *text, const smh_wm_params &P, bool cond, uint64_t e)
{
    const uint64_t mask = __ballot(cond);
    if (mask == 0) return;
    const uint32_t np = (uint32_t)__popcll(mask);
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (cond) Q.slots[Q.count + before] = e; /* LDS, written and read by this wave only */
    Q.count += np;
    Q.events += cond ? 1u : 0u;
}
#else
SMH_LANE void smh_wm_drain(smh_wm_queue &, const uint8_t *, const smh_wm_params &) {}
SMH_LANE void smh_wm_emit(smh_wm_queue &Q, const uint8_t *text, const smh_wm_params &P, bool cond, uint64_t e)
{
    if (!cond) return;
    if (P.n_classes && P.sfx_slot && e >= 31u) {
        Q.matches += smh_wm_verify_sfx(text, e, true, P, Q.po);
        return;
    }
    if (P.n_classes) {
        for (int c = 0; c < P.n_classes; ++c) {
            const uint32_t hit = smh_wm_verify_class(text, e, P, c);
            Q.matches += hit;
            if (hit && Q.po) smh_append_bits(1u, e, *Q.po);
        }
        return;
    }
    const uint32_t hit = smh_wm_verify(text, e, P);
    Q.matches += hit;
    if (hit && Q.po) smh_append_bits(1u, e, *Q.po);
}
#endif

/* ---- the software-pipelined probe of the IN-REGISTER verify (smh_wm_regv_columns): one pending column per lane, its first
 * bucket requested between chunks and looked at after the next chunk's scan.  Compiled for the CPU emulation too (round 4):
 * the emulator runs a lane at a time, so "the lanes that hold a column" is simply "this lane", and it walks the very same
 * issue / finish / carry-over-to-the-next-chunk control flow as the GPU. */
SMH_LANE void smh_wm_pend_issue_rv(smh_wm_queue &Q, const smh_wm_params &P)
{
    if (Q.pend_n == 0 || Q.pend_loaded) return;
    if (Q.pend_mine) Q.pend_q = smh_load16(reinterpret_cast<const uint8_t *>(P.verify) + 16u * (uint64_t)smh_wm_first_bucket(Q.pend_tag, P));
    Q.pend_loaded = 1u;
}
SMH_LANE void smh_wm_pend_finish_rv(smh_wm_queue &Q, const uint8_t *text, const smh_wm_params &P)
{
    if (Q.pend_n == 0) return;
    uint32_t r = 0;
    if (Q.pend_mine) r = smh_wm_probe_from(text, Q.pend_e, Q.pend_tag, P, Q.pend_loaded != 0u, Q.pend_q);
    Q.matches += r;
    if (Q.po) smh_append_bits(r, Q.pend_e, *Q.po);
    Q.pend_mine = 0u;
    Q.pend_n = 0u;
    Q.pend_loaded = 0u;
}

/* ---- staged verify, wave level (description above SMH_STAGE_BUF) ---- */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
typedef uint32_t smh_lds_v4 __attribute__((ext_vector_type(4)));
SMH_LANE void smh_lds_store16(uint32_t byte_off, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    *reinterpret_cast<__attribute__((address_space(3))) smh_lds_v4 *>(byte_off) = smh_lds_v4{a, b, c, d};
}

/* verify the queue entries [from, Q.count) -- all columns of the wave-chunk at `chunk_base`, whose text the lanes
 * hold in w (and lane 0 the 16*STG bytes in front of it in `halo`) -- and remove them; at most 128 entries; all 64
 * lanes must call it */
/* request the first bucket of the pending columns (call between chunks, before the next chunk's text is requested) */
template <bool RV = false>
SMH_LANE void smh_wm_pend_issue(smh_wm_queue &Q, const smh_wm_params &P)
{
    if constexpr (RV) { /* only the lanes that hold a column ask for a bucket */
        smh_wm_pend_issue_rv(Q, P);
        return;
    }
    if (Q.pend_n == 0 || Q.pend_loaded) return;
    if (P.verify_ck) Q.pend_q = smh_wm_ck_load(Q.pend_tag, P); /* wave-uniform */
    else Q.pend_q = smh_load16(reinterpret_cast<const uint8_t *>(P.verify) + 16u * (uint64_t)smh_wm_first_bucket(Q.pend_tag, P));
    Q.pend_loaded = 1u;
}
/* decide the pending columns; all 64 lanes must call it */
template <bool RV = false>
SMH_LANE void smh_wm_pend_finish(smh_wm_queue &Q, const uint8_t *text, const smh_wm_params &P)
{
    if constexpr (RV) {
        smh_wm_pend_finish_rv(Q, text, P);
        return;
    }
    if (Q.pend_n == 0) return;
    const bool mine = (threadIdx.x & 63u) < Q.pend_n;
    const uint32_t r = P.verify_ck ? smh_wm_ck_decide(text, Q.pend_e, Q.pend_tag, P, Q.pend_loaded != 0u ? Q.pend_q : smh_wm_ck_load(Q.pend_tag, P))
                                   : smh_wm_probe_from(text, Q.pend_e, Q.pend_tag, P, Q.pend_loaded != 0u, Q.pend_q);
    Q.matches += mine ? r : 0u;
    if (Q.po) smh_append_bits(mine ? r : 0u, Q.pend_e, *Q.po);
    Q.pend_n = 0u;
    Q.pend_loaded = 0u;
}

/* PIPE: the first 64 columns are left pending (smh_wm_pend_*) instead of being probed at once */
template <int STG, bool QD = true, bool PIPE = false>
SMH_LANE void smh_wm_stage_flush(smh_wm_queue &Q, const uint8_t *text, uint64_t chunk_base, uint32_t from,
                                 const uint32_t (&w)[16], const uint32_t (&halo)[4 * STG], const smh_wm_params &P)
{
    if (Q.count <= from) return;
    const uint32_t lane = threadIdx.x & 63u, cnt = Q.count - from;
    constexpr uint32_t HALO = 16u * STG, BUF = SMH_STAGE_BUF(STG);
    /* take a staging buffer: lane 0 tries the lock words in turn, starting at a wave-dependent one */
    uint32_t i = (threadIdx.x >> 6) % Q.st_nbufs;
    for (;;) {
        uint32_t got = 0;
        if (lane == 0) {
            uint32_t expected = 0u;
            got = __hip_atomic_compare_exchange_strong(
                      reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(Q.st_locks + 4u * i), &expected, 1u,
                      __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
        }
        if (__builtin_amdgcn_readfirstlane((int)got)) break;
        i = i + 1u == Q.st_nbufs ? 0u : i + 1u;
        __builtin_amdgcn_s_sleep(2);
    }
    const uint32_t buf = Q.st_bufs + i * BUF;
#pragma unroll
    for (int q = 0; q < 4; ++q) smh_lds_store16(buf + HALO + lane * 64u + 16u * q, w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    if (lane == 0) { /* lane 0's view of the bytes in front of the chunk (wave-uniform in the gram kernels) */
#pragma unroll
        for (int q = 0; q < STG; ++q) smh_lds_store16(buf + 16u * q, halo[4 * q], halo[4 * q + 1], halo[4 * q + 2], halo[4 * q + 3]);
    }
    /* lane l hashes the windows of entries l and l + 64 */
    const bool h0 = lane < cnt, h1 = lane + 64u < cnt;
    /* QD: 64-bit END columns (entries of other chunks may wait in front of them); else 32-bit chunk offsets */
    auto entry = [&](uint32_t i) -> uint64_t {
        if constexpr (QD) return Q.slots[from + i];
        else return chunk_base + reinterpret_cast<const uint32_t *>(Q.slots)[i];
    };
    const uint64_t e0 = entry(h0 ? lane : 0u);
    auto rd = [&](uint32_t off) { return smh_lds_u32(nullptr, buf + off); };
    const uint32_t tag0 = smh_wm_tag_staged<SMH_STAGE_MAXD(STG)>(rd, (uint32_t)(e0 - chunk_base) + HALO + 1u - (uint32_t)P.m, P.m);
    uint64_t e1 = e0;
    uint32_t tag1 = tag0;
    if (cnt > 64u) { /* wave-uniform */
        e1 = entry(h1 ? lane + 64u : 0u);
        tag1 = smh_wm_tag_staged<SMH_STAGE_MAXD(STG)>(rd, (uint32_t)(e1 - chunk_base) + HALO + 1u - (uint32_t)P.m, P.m);
    }
    /* hand the buffer back: the release waits for the window reads above, nothing else */
    if (lane == 0)
        __hip_atomic_store(reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(Q.st_locks + 4u * i), 0u,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if constexpr (PIPE) {
        smh_wm_pend_finish(Q, text, P); /* the chunk before: its bucket arrived while this one was scanned */
        Q.pend_n = cnt < 64u ? cnt : 64u;
        Q.pend_e = e0;
        Q.pend_tag = tag0;
        if (cnt > 64u) { /* the overflow of a dense chunk is decided at once */
            const uint32_t r1 = smh_wm_probe(text, e1, tag1, P);
            Q.matches += h1 ? r1 : 0u;
            if (Q.po) smh_append_bits(h1 ? r1 : 0u, e1, *Q.po);
        }
    } else {
    uint32_t r0, r1 = 0;
    if (cnt > 64u)
        r0 = smh_wm_probe2(text, e0, e1, tag0, tag1, P, r1);
    else
        r0 = smh_wm_probe(text, e0, tag0, P);
    Q.matches += (h0 ? r0 : 0u) + (h1 ? r1 : 0u);
    if (Q.po) {
        smh_append_bits(h0 ? r0 : 0u, e0, *Q.po);
        if (cnt > 64u) smh_append_bits(h1 ? r1 : 0u, e1, *Q.po);
    }
    }
    Q.count = from;
}

/* the surviving columns `msk` (bit b = column a + b) of a lane's segment in the wave-chunk at chunk_base: queue
 * them; a chunk with st_min or more of them is verified from its staged copy at once, the others wait in the
 * queue for the next drain from HBM */
/* QD = the kernel also has the drain from HBM (sparse chunks wait in the queue).  Without it every chunk with a
 * surviving column is staged: the drain's code inside the chunk loop costs the scan 10 % even when it never runs
 * (gpurun_out/r02_v: 0.175 -> 0.193 ms/GiB on a set without survivors), so the launcher picks it only for sets
 * that expect a few survivors per chunk. */
template <int STG, bool QD = true, bool PIPE = false>
SMH_LANE void smh_wm_stage_columns(smh_wm_queue &Q, const uint8_t *text, uint64_t chunk_base, uint64_t a, uint64_t msk,
                                   const uint32_t (&w)[16], const uint32_t (&halo)[4 * STG], const smh_wm_params &P)
{
    if (!SMH_WAVE_ANY(msk != 0)) return;
    if (Q.st_min == 0xFFFFFFFFu) return; /* development knob SMH_WM_TUNE="stmin=-1": survivors are dropped (what the bare filter scan costs; counts are wrong) */
    uint32_t from = QD ? Q.count : 0u; /* entries of earlier chunks */
    do {
        if (Q.count + 64u > SMH_WM_QCAP) {
            smh_wm_stage_flush<STG, QD, PIPE>(Q, text, chunk_base, from, w, halo, P);
            if constexpr (QD) {
                if (Q.count + 64u > SMH_WM_QCAP) { /* still full: the earlier chunks' entries, from HBM */
                    smh_wm_drain(Q, text, P);
                    from = 0;
                }
            }
        }
        const bool have = msk != 0;
        const uint32_t b = have ? (uint32_t)__builtin_ctzll(msk) : 0u;
        if constexpr (QD) {
            smh_wm_emit(Q, text, P, have, a + b);
        } else { /* the queue only ever holds this chunk's columns: 32-bit offsets, half the LDS */
            const uint64_t mask = __ballot(have);
            const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            if (have) reinterpret_cast<uint32_t *>(Q.slots)[Q.count + before] = (uint32_t)(a - chunk_base) + b;
            Q.count += (uint32_t)__popcll(mask);
            Q.events += have ? 1u : 0u;
        }
        msk &= msk - 1u;
    } while (SMH_WAVE_ANY(msk != 0));
    if (!QD || Q.count - from >= Q.st_min) smh_wm_stage_flush<STG, QD, PIPE>(Q, text, chunk_base, from, w, halo, P);
}
#else
/* CPU emulation (one lane at a time): the same window hash over a private copy of the chunk laid out as the
 * staging buffer, the same probe */
template <int STG>
SMH_LANE void smh_wm_stage_verify_emu(smh_wm_queue &Q, const uint8_t *text, uint64_t chunk_base, uint32_t c, const smh_wm_params &P)
{
    constexpr uint32_t HALO = 16u * STG;
    uint8_t buf[SMH_STAGE_BUF(STG)];
    memset(buf, 0xA5, sizeof buf); /* the pad is never part of a hash */
    memcpy(buf, text + chunk_base - HALO, HALO + 4096u);
    auto rd = [&](uint32_t off) { uint32_t v; memcpy(&v, buf + off, 4); return v; };
    const uint32_t tag = smh_wm_tag_staged<SMH_STAGE_MAXD(STG)>(rd, c + HALO + 1u - (uint32_t)P.m, P.m);
    const uint32_t hit = smh_wm_probe(text, chunk_base + c, tag, P);
    Q.matches += hit;
    if (hit && Q.po) smh_append_bits(1u, chunk_base + c, *Q.po);
}
template <int STG, bool QD = true, bool PIPE = false>
SMH_LANE void smh_wm_stage_columns(smh_wm_queue &Q, const uint8_t *text, uint64_t chunk_base, uint64_t a, uint64_t msk,
                                   const uint32_t (&)[16], const uint32_t (&)[4 * STG], const smh_wm_params &P)
{
    while (msk) {
        smh_wm_stage_verify_emu<STG>(Q, text, chunk_base, (uint32_t)(a - chunk_base) + (uint32_t)__builtin_ctzll(msk), P);
        msk &= msk - 1u;
    }
}
#endif

/* ---- windows from L2 (STG 3 / 4; description at smh_stg_l2) ----
 * the aligned dwords of the m-byte window that ends at column e, requested in one go; dwords the window does not reach
 * are not touched (the text buffer ends with its last byte's dword) */
template <int MAXD>
SMH_LANE uint32_t smh_wm_l2_request(const uint8_t *text, uint64_t e, int m, uint32_t (&d)[MAXD + 1], bool wide)
{
    const uint64_t s0 = e + 1 - (uint64_t)m;
    const uint32_t *aligned = reinterpret_cast<const uint32_t *>(text + (s0 & ~(uint64_t)3));
    const uint32_t sh = (uint32_t)(s0 & 3u) * 8u;
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    if (MAXD == 9 && wide) {
        /* wave-uniform: 40 bytes from the window's first dword on lie inside the text.  The long window as two 16-byte
         * requests and an 8-byte one (dword-aligned) instead of ten 4-byte ones: 100 000 patterns of 20 bytes 1.21 -> 1.16 ms
         * per 4 GiB.  (The short window -- six 4-byte requests, most of them bent onto its last dword -- is faster as it is:
         * one 16-byte + one 8-byte request measured 1.13 -> 1.20 ms at 12 bytes; gpurun_out/r04_p/ab_l2wide.log.) */
        typedef uint32_t v4a __attribute__((ext_vector_type(4), aligned(4)));
        typedef uint32_t v2a __attribute__((ext_vector_type(2), aligned(4)));
        const v4a q0 = *reinterpret_cast<const v4a *>(aligned);
        d[0] = q0.x; d[1] = q0.y; d[2] = q0.z; d[3] = q0.w;
        const v4a q1 = *reinterpret_cast<const v4a *>(aligned + 4);
        d[(MAXD + 1) / 2 - 1] = q1.x; d[(MAXD + 1) / 2] = q1.y; d[(MAXD + 1) / 2 + 1] = q1.z; d[(MAXD + 1) / 2 + 2] = q1.w;
        const v2a q2 = *reinterpret_cast<const v2a *>(aligned + (MAXD - 1));
        d[MAXD - 1] = q2.x; d[MAXD] = q2.y;
        return sh;
    }
#else
    (void)wide;
#endif
    const uint32_t last = (sh + 8u * (uint32_t)m - 1u) >> 5; /* index of the dword that holds the window's last byte */
    /* a dword beyond it is never part of the hash (smh_wm_tag_dwords masks the last dword to the window): the request is
     * bent back onto dword `last` instead of predicated away */
#pragma unroll
    for (int j = 0; j <= MAXD; ++j) d[j] = aligned[(uint32_t)j < last ? (uint32_t)j : last];
    return sh;
}

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
/* the pipeline's step at a chunk boundary: decide the columns whose bucket has arrived, hash the windows that have arrived
 * and request their buckets, take up to 64 new columns from the queue and request their windows.  All 64 lanes call it. */
template <int MAXD>
SMH_LANE void smh_wm_l2_step(smh_wm_queue &Q, const uint8_t *text, uint64_t chunk_base, const smh_wm_params &P, bool take)
{
    Q.pend_mine = 1u; /* (a field of the in-register verify, free here: "the pipeline has moved in this iteration") */
    smh_wm_pend_finish<false>(Q, text, P);
    if (Q.pa_n) {
        uint32_t d[MAXD + 1];
#pragma unroll
        for (int j = 0; j <= MAXD; ++j) d[j] = Q.pa_w[j];
        Q.pend_tag = smh_wm_tag_dwords<MAXD>(d, Q.pa_sh, P.m);
        Q.pend_e = Q.pa_e;
        Q.pend_n = Q.pa_n;
        Q.pend_loaded = 0u;
        Q.pa_n = 0u;
        smh_wm_pend_issue<false>(Q, P);
    }
    if (!take || Q.count == 0) return;
    /* the queue holds END columns (64-bit), of this chunk and of earlier ones (smh_wm_l2_columns): the LAST 64 of them are
     * taken, one per lane; what lies in front of them stays for the next step */
    const uint32_t lane = threadIdx.x & 63u, cnt = Q.count;
    const uint32_t n_take = cnt < 64u ? cnt : 64u, first = cnt - n_take;
    /* a lane without an entry re-requests the first one taken (a valid address; its answer is dropped) */
    Q.pa_e = Q.slots[first + (lane < n_take ? lane : 0u)];
    uint32_t d[MAXD + 1];
    /* wave-uniform: 40 bytes from every window's first dword on lie inside the text (the first dword is at or below the column) */
    const bool wide = MAXD == 9 && !SMH_WAVE_ANY(Q.pa_e + 44u > Q.pa_limit);
    Q.pa_sh = smh_wm_l2_request<MAXD>(text, Q.pa_e, P.m, d, wide);
#pragma unroll
    for (int j = 0; j <= MAXD; ++j) Q.pa_w[j] = d[j];
    Q.pa_n = n_take;
    Q.count = first;
}

/* the surviving columns `msk` (bit b = column a + b) of a lane's segment in the wave-chunk at chunk_base */
template <int MAXD>
SMH_LANE void smh_wm_l2_columns(smh_wm_queue &Q, const uint8_t *text, uint64_t chunk_base, uint64_t a, uint64_t msk, const smh_wm_params &P)
{
    if (!SMH_WAVE_ANY(msk != 0)) return;
    if (Q.st_min == 0xFFFFFFFFu) return; /* development knob: survivors dropped (smh_wm_stage_columns) */
    do {
        if (Q.count + 64u > SMH_WM_QCAP) smh_wm_l2_step<MAXD>(Q, text, chunk_base, P, true); /* more than 64 queued: take 64 now */
        const bool have = msk != 0;
        const uint32_t b = have ? (uint32_t)__builtin_ctzll(msk) : 0u;
        const uint64_t mask = __ballot(have);
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (have) Q.slots[Q.count + before] = a + b;
        Q.count += (uint32_t)__popcll(mask);
        Q.events += have ? 1u : 0u;
        msk &= msk - 1u;
    } while (SMH_WAVE_ANY(msk != 0));
    /* Round 5: a step costs the wave the same instructions whether 27 of its lanes hold a column or 64 (window request, hash,
     * bucket request, four tag compares: about a quarter of the kernel's vector instructions at 27 survivors per chunk, taken
     * after every chunk).  So the columns wait in the queue -- as END columns, not chunk offsets -- until st_min (48) of them
     * are there, about every second chunk on uniform text; a chunk that takes nothing still moves the pipeline on
     * (smh_wm_gram_thread), and the end of the wave's work empties the queue.  100 000 byte patterns, 4 GiB: m = 8 1.249 -> 1.207 ms,
     * m = 20 1.203 -> 1.141, m = 12 and 5 within 1 %; vector instructions per column 6.45 -> 5.80 (profiles/r05_final/notes/
     * ab_verify_pipeline.log, which also holds what did NOT pay: one 16-byte window request instead of six 4-byte ones, and the
     * next chunk's text prefetched in front of the step -- the cost of a survivor is its two scattered lines, not their order). */
    if (Q.st_min == 0xFFFFFFFEu) { Q.count = 0; return; } /* development knob "stmin=-2": survivors queued, then dropped */
    if (Q.count >= Q.st_min) smh_wm_l2_step<MAXD>(Q, text, chunk_base, P, true);
}
#else
/* CPU emulation (one lane at a time): the same window request and hash, decided at once */
template <int MAXD>
SMH_LANE void smh_wm_l2_columns(smh_wm_queue &Q, const uint8_t *text, uint64_t, uint64_t a, uint64_t msk, const smh_wm_params &P)
{
    while (msk) {
        const uint64_t e = a + (uint64_t)__builtin_ctzll(msk);
        uint32_t d[MAXD + 1];
        const uint32_t sh = smh_wm_l2_request<MAXD>(text, e, P.m, d, false);
        const uint32_t tag = smh_wm_tag_dwords<MAXD>(d, sh, P.m);
        const uint32_t hit = P.verify_ck ? smh_wm_ck_decide(text, e, tag, P, smh_wm_ck_load(tag, P)) : smh_wm_probe(text, e, tag, P);
        Q.matches += hit;
        if (hit && Q.po) smh_append_bits(1u, e, *Q.po);
        msk &= msk - 1u;
    }
}
#endif

/*
 * Fast path: the lane owns the 64 END columns of the segment at byte offset a
 * (a multiple of 64) and reads the 16*HC bytes in front of it to prime the
 * rolling block code; the caller guarantees a >= 16*HC, a + 64 <= n and
 * 16*HC >= m-1, so every column has a full window.
 */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE uint32_t smh_alignbyte(uint32_t hi, uint32_t lo, uint32_t r) { return __builtin_amdgcn_alignbyte(hi, lo, r); }
#else
SMH_LANE uint32_t smh_alignbyte(uint32_t hi, uint32_t lo, uint32_t r)
{
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8u * r));
}
#endif

/* FK > 0: the byte-symbol tuned path -- the block is the column's last four bytes, read as one
 * unaligned little-endian dword out of the lane's registers (v_alignbyte), hashed filter with FK
 * bits per key.  FK == 0: any symbol width, rolling code. */
template <bool HASHED, bool EXACT, int HC, int FK = 0, bool POS = false, bool STG = false>
SMH_LANE uint32_t smh_wm_lane_fast(const uint8_t *text, uint64_t a, const uint32_t (&w)[4 * HC + 16],
                                   const uint32_t *filter, const smh_wm_params &P, smh_wm_queue &Q)
{
    uint32_t code = 0, cnt = 0;
    uint32_t surv[2] = {0, 0};
    if constexpr (FK > 0) {
        static_assert(HASHED && !EXACT, "the byte-block path is the hashed, verified one");
        auto key_of = [&](int i) {
            const int first = 16 * HC + i - 3; /* register byte index of the block's first byte */
            const int d = first >> 2, r = first & 3;
            return r == 0 ? w[d] : smh_alignbyte(w[d + 1], w[d], (uint32_t)r);
        };
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
        /* four columns at a time: addresses, then the four block reads in flight together, then the tests.  Bit 0 of a
         * test's result goes into the survivor mask at the top and the earlier columns move down (v_alignbit): after 32
         * columns bit i is column i's */
        typedef uint32_t smh_v2u __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i0 = 0; i0 < 64; i0 += 4) {
            uint32_t g[4], hh[4], addr[4];
            smh_v2u blk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) addr[j] = smh_wm_v2_addr(key_of(i0 + j), g[j], hh[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) blk[j] = *reinterpret_cast<const __attribute__((address_space(3))) smh_v2u *>(addr[j]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                surv[i0 >> 5] = __builtin_amdgcn_alignbit(smh_wm_v2_test<FK>(g[j], hh[j], blk[j].x, blk[j].y), surv[i0 >> 5], 1u);
        }
#else
#pragma unroll
        for (int i = 0; i < 64; ++i) surv[i >> 5] |= smh_wm_filter_key_v2<FK>(key_of(i), filter, P) << (i & 31);
#endif
    } else {
#pragma unroll
    for (int i = 0; i < 16 * HC; ++i) code = (code << P.bits) | smh_byte_of(w[i >> 2], i & 3);
    /* SHIFT stage over the 64 columns; survivors are only recorded (one bit each) so that the
     * unrolled loop stays branch-free */
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        code = (code << P.bits) | smh_byte_of(w[4 * HC + (i >> 2)], i & 3);
        const uint32_t hit = smh_wm_filter<HASHED>(code, filter, P);
        if (EXACT && !POS)
            cnt += hit;
        else
            surv[i >> 5] |= hit << (i & 31);
    }
    }
    if (EXACT && POS) /* a set filter bit IS a match: append the END columns */
        cnt += smh_append_bits(((uint64_t)surv[1] << 32) | surv[0], a, *Q.po);
    if (!EXACT) {
        /* compaction: one queue entry per surviving column, as many rounds as the busiest lane has;
         * the HASH/PREFIX stage (drain) is entered from this one place while the wave scans */
        uint64_t msk = ((uint64_t)surv[1] << 32) | surv[0];
        if constexpr (STG) {
            /* staged verify (see smh_wm_stage_flush): the 16 * HC bytes each lane loaded in front of its segment are,
             * in lane 0, the bytes in front of the wave-chunk */
            static_assert(!STG || (HC >= 1 && HC <= 2), "staged verify covers a halo of 16 or 32 bytes");
            uint32_t own[16], halo[4 * HC];
#pragma unroll
            for (int q = 0; q < 16; ++q) own[q] = w[4 * HC + q];
#pragma unroll
            for (int q = 0; q < 4 * HC; ++q) halo[q] = w[q];
            smh_wm_stage_columns<HC>(Q, text, smh_uniform64(a & ~(uint64_t)4095), a, msk, own, halo, P);
        } else {
        while (SMH_WAVE_ANY(msk != 0)) {
            if (Q.count + 64u > SMH_WM_QCAP) smh_wm_drain(Q, text, P);
            const bool have = msk != 0;
            const uint32_t b = have ? (uint32_t)__builtin_ctzll(msk) : 0u;
            smh_wm_emit(Q, text, P, have, a + b);
            msk &= msk - 1u;
        }
        }
    }
    return cnt;
}

/* Slow path: any segment of END columns, bounds checked, no pre-halo requirement. */
template <bool HASHED, bool EXACT>
SMH_LANE uint32_t smh_wm_lane_slow(const uint8_t *text, uint64_t n, uint64_t a, const uint32_t *filter,
                                   const smh_wm_params &P, int block_symbols, uint64_t *match_mask = nullptr,
                                   int only_class = -1)
{
    if (match_mask) *match_mask = 0;
    if (a >= n) return 0;
    uint64_t end = a + SMH_SEG;
    if (end > n) end = n;
    uint64_t e0 = a;
    if (e0 < (uint64_t)(P.m - 1)) e0 = (uint64_t)(P.m - 1);
    if (e0 >= end) return 0;
    /* prime with the block_symbols - 1 symbols in front of the first column (they exist: e0 >= m-1) */
    uint32_t code = 0, cnt = 0;
    for (uint64_t i = e0 - (uint64_t)(block_symbols - 1); i < e0; ++i) code = (code << P.bits) | text[i];
    for (uint64_t e = e0; e < end; ++e) {
        code = (code << P.bits) | text[e];
        uint32_t hit = smh_wm_filter<HASHED>(code, filter, P);
        if (!EXACT && hit) hit = only_class >= 0 ? smh_wm_verify_class(text, e, P, only_class) : smh_wm_verify_any(text, e, P);
        cnt += hit;
        if (match_mask && hit) *match_mask |= 1ull << (e - a); /* positions mode: bit = END column - a */
    }
    return cnt;
}

/*
 * SMH_VARIANT_TABLE: the reference's loop (wu/wu.c:61-104, cuda/cuda_wm.cu:1012-1057)
 * over the END columns [a, a + span) with the reference tables as given:
 * shift[] (LDS copy of SHIFT), CSR buckets of {PREFIX_value, PREFIX_index}.
 */
template <typename SHIFT_T>
SMH_LANE uint32_t smh_wm_lane_table(const uint8_t *text, uint64_t n, uint64_t a, uint64_t span,
                                    const SHIFT_T *shift, uint32_t shiftsize, const uint32_t *bucket_off,
                                    const int32_t *bucket, const uint8_t *pat_orig, int m, int nbits)
{
    uint64_t end = a + span;
    if (end > n) end = n;
    uint64_t column = a;
    if (column < (uint64_t)(m - 1)) column = (uint64_t)(m - 1);
    uint32_t cnt = 0;
    while (column < end) {
        uint32_t hash1 = text[column - 2];
        hash1 <<= nbits;
        hash1 += text[column - 1];
        hash1 <<= nbits;
        hash1 += text[column];
        /* a byte >= alphabet can push hash1 past the table; such a column ends no match */
        const uint32_t sh = hash1 < shiftsize ? shift[hash1] : 1u;
        if (sh == 0) {
            uint32_t hash2 = text[column - (uint64_t)m + 1];
            hash2 <<= nbits;
            hash2 += text[column - (uint64_t)m + 2];
            const uint32_t b0 = bucket_off[hash1], b1 = bucket_off[hash1 + 1];
            for (uint32_t k = b0; k < b1; ++k) {
                if ((uint32_t)bucket[2 * k] != hash2) continue;
                const uint8_t *q = pat_orig + (uint64_t)(uint32_t)bucket[2 * k + 1] * (uint32_t)m;
                const uint8_t *w = text + (column + 1 - (uint64_t)m);
                int i = 0;
                while (i < m && q[i] == w[i]) ++i;
                if (i == m) {
                    ++cnt;
                    break;
                }
            }
            ++column;
        } else {
            column += sh;
        }
    }
    return cnt;
}

/* whole-grid work distribution for one lane; HC == 0: no fast path (m - 1 > 64).  The next chunk's
 * text is requested before the current chunk is scanned (software prefetch). */
template <bool HASHED, bool EXACT, int HC, int FK = 0, bool POS = false, bool STG = false>
SMH_LANE uint32_t smh_wm_thread(uint64_t gthread, const smh_chunk_sched &S, const uint8_t *text, uint64_t n,
                                const uint32_t *filter, const smh_wm_params &P, int block_symbols,
                                uint64_t *queue_base, const smh_pos_out *po = nullptr, const smh_wm_queue *stage = nullptr,
                                uint32_t *events_out = nullptr)
{
    if (n < (uint64_t)P.m) return 0;
    constexpr int H = HC > 0 ? HC : 1;
    const uint64_t chunk_bytes = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (n + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    smh_wm_queue Q = {};
    if (stage) Q = *stage; /* staged verify (STG): where the locks, buffers and this wave's list live */
    Q.slots = queue_base; /* this wave's slice (the kernel passes LDS) */
    Q.count = 0;
    Q.matches = 0;
    Q.events = 0;
    Q.po = POS ? po : nullptr;
    uint32_t cnt = 0;
    uint32_t cur[4 * H + 16], nxt[4 * H + 16];
    auto is_fast = [&](uint64_t kk) {
        return HC > 0 && kk < n_chunks && kk * chunk_bytes >= 16u * H && (kk + 1) * chunk_bytes <= n;
    };
    auto load = [&](uint64_t kk, uint32_t (&w)[4 * H + 16]) {
        const uint8_t *p = text + smh_uniform64(kk * chunk_bytes) + (uint64_t)lane * SMH_SEG - 16u * H;
#pragma unroll
        for (int q = 0; q < H + 4; ++q) {
            const smh_u32x4 t = smh_load16(p + 16u * q);
            w[4 * q + 0] = t.v[0];
            w[4 * q + 1] = t.v[1];
            w[4 * q + 2] = t.v[2];
            w[4 * q + 3] = t.v[3];
        }
    };
    uint64_t k = S.take(n_chunks);
    bool cur_fast = is_fast(k);
    if (cur_fast) load(k, cur);
    while (k < n_chunks) {
        const uint64_t kn = S.take(n_chunks);
        const bool nxt_fast = is_fast(kn);
        constexpr bool PREFETCH = SMH_PREFETCH && EXACT && H == 1; /* lane_common.h; only where registers allow: exact filter, short pre-halo */
        if (STG && Q.count >= 64u) smh_wm_drain(Q, text, P); /* columns of sparse chunks, from HBM, 64 at a time */
        if (PREFETCH && nxt_fast) load(kn, nxt);
        const uint64_t a = smh_uniform64(k * chunk_bytes) + (uint64_t)lane * SMH_SEG;
        if (cur_fast) {
            cnt += smh_wm_lane_fast<HASHED, EXACT, H, FK, POS, STG>(text, a, cur, filter, P, Q);
        } else if (POS) {
            /* first / last chunks: per-lane mask of matching END columns, then the wave-level append */
            uint64_t mm;
            if (P.n_classes) { /* a column is appended once per length class that matches there */
                for (int c = 0; c < P.n_classes; ++c) {
                    smh_wm_lane_slow<HASHED, EXACT>(text, n, a, filter, P, block_symbols, &mm, c);
                    cnt += smh_append_bits(mm, a, *po);
                }
            } else {
                smh_wm_lane_slow<HASHED, EXACT>(text, n, a, filter, P, block_symbols, &mm);
                cnt += smh_append_bits(mm, a, *po);
            }
        } else {
            cnt += smh_wm_lane_slow<HASHED, EXACT>(text, n, a, filter, P, block_symbols);
        }
        if (nxt_fast) {
            if (PREFETCH) {
#pragma unroll
                for (int q = 0; q < 4 * H + 16; ++q) cur[q] = nxt[q];
            } else {
                load(kn, cur);
            }
        }
        cur_fast = nxt_fast;
        k = kn;
    }
    if (!EXACT) smh_wm_drain(Q, text, P);
    if (events_out) *events_out = Q.events;
    return cnt + Q.matches;
}

#define SMH_WM_TABLE_SPAN 256u /* END columns per lane in the table-walking kernel */
template <typename SHIFT_T>
SMH_LANE uint32_t smh_wm_table_thread(uint64_t gthread, uint64_t nthreads, const uint8_t *text, uint64_t n,
                                      const SHIFT_T *shift, uint32_t shiftsize, const uint32_t *bucket_off,
                                      const int32_t *bucket, const uint8_t *pat_orig, int m, int nbits)
{
    if (n < (uint64_t)m) return 0;
    uint32_t cnt = 0;
    for (uint64_t a = gthread * SMH_WM_TABLE_SPAN; a < n; a += nthreads * SMH_WM_TABLE_SPAN)
        cnt += smh_wm_lane_table<SHIFT_T>(text, n, a, SMH_WM_TABLE_SPAN, shift, shiftsize, bucket_off, bucket,
                                          pat_orig, m, nbits);
    return cnt;
}

/* ------------------------------------------------------------------ pair filter (alphabet 4, m <= 8)
 * One LDS lookup decides TWO end columns: the lookup key is the 18-bit code of the nine symbols
 * ending at the second column (smh_internal.h "pair filter").  Per pair of text bytes: one
 * v_bfe (pair code), one v_lshl_or (rolling code), two ops for the address, one ds_read_b64,
 * two v_bfe with the code's low 5 bits as bit index, one v_add3.  The eight symbols in front of
 * the segment come from the previous lane's registers (DPP wave_shr:1); lane 0 takes the
 * wave-uniform `edge` words loaded from the 8 bytes in front of the wave-chunk.
 */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE uint32_t smh_prev_lane_word(uint32_t mine, uint32_t edge, const uint8_t *, uint64_t)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)mine, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
SMH_LANE void smh_lds_u32x2(const void *, uint32_t byte_off, uint32_t &lo, uint32_t &hi)
{
    typedef uint32_t smh_v2u __attribute__((ext_vector_type(2)));
    const smh_v2u v = *reinterpret_cast<const __attribute__((address_space(3))) smh_v2u *>(byte_off);
    lo = v.x;
    hi = v.y;
}
#else
SMH_LANE uint32_t smh_prev_lane_word(uint32_t, uint32_t, const uint8_t *text, uint64_t byte_offset)
{
    uint32_t v;
    memcpy(&v, text + byte_offset, 4);
    return v;
}
SMH_LANE void smh_lds_u32x2(const void *base, uint32_t byte_off, uint32_t &lo, uint32_t &hi)
{
    memcpy(&lo, (const uint8_t *)base + byte_off, 4);
    memcpy(&hi, (const uint8_t *)base + byte_off + 4, 4);
}
#endif

/* Pair filter lookups (table layout: smh_internal.h).  x = (w << 10) | w puts the pair codes (4 bits, first
 * symbol high) of a text dword at bits 8..11 and 24..27 with zero bits below them, so that one v_bfe yields the
 * code times 2 or times 4.  `c2` is the rolling code of the symbols seen so far, times 4: its low 16 bits ARE
 * the byte address of the dword that the seven symbols before the new pair select.  Per pair of columns:
 * v_and (address), ds_read_b32, v_bfe (pair * 4), v_lshl_or (roll), v_bfe (pair * 2), v_bfe (the two answers),
 * v_bcnt (count). */
SMH_LANE uint32_t smh_wm_pairs_prep(uint32_t w) { return (w << 10) | w; }

/* the two answers for the pair k of x as two bits (bit 0 = the pair's first column, bit 1 = its second) */
SMH_LANE uint32_t smh_wm_pair_step_bits(uint32_t &c2, uint32_t x, int k, const void *tab)
{
    const uint32_t word = smh_lds_u32(tab, c2 & 0xFFFCu);
    c2 = (c2 << 4) | smh_bfe(x, k == 0 ? 6 : 22, 6);
    return smh_bfe(word, smh_bfe(x, k == 0 ? 7 : 23, 5), 2);
}

/* fast path: the 64 END columns of the segment at a (a >= 8, a + 64 <= n, first column >= m-1) */
template <bool POS = false>
SMH_LANE uint32_t smh_wm_pair_lane_fast(const uint8_t *text, uint64_t a, const uint32_t (&w)[16],
                                        const uint32_t (&edge)[2], const void *tab, const smh_pos_out *po = nullptr)
{
    /* prime the rolling code with the 8 symbols in front of the segment */
    uint32_t c2 = 0, cnt = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const uint32_t pw = smh_prev_lane_word(w[14 + q], edge[q], text, a - 8u + 4u * q);
        const uint32_t x = smh_wm_pairs_prep(pw);
        c2 = (c2 << 4) | smh_bfe(x, 6, 6);
        c2 = (c2 << 4) | smh_bfe(x, 22, 6);
    }
    /* `c2` now holds symbols a-8 .. a-1; the lookup for the pair (a+2i, a+2i+1) sees a+2i-7 .. a+2i+1 */
    if constexpr (POS) {
        uint32_t mlo = 0, mhi = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t x = smh_wm_pairs_prep(w[q]);
            const uint32_t b0 = smh_wm_pair_step_bits(c2, x, 0, tab), b1 = smh_wm_pair_step_bits(c2, x, 1, tab);
            const uint32_t four = b0 | (b1 << 2);
            if (q < 8)
                mlo |= four << (4 * q);
            else
                mhi |= four << (4 * (q - 8));
        }
        return smh_append_bits(((uint64_t)mhi << 32) | mlo, a, *po);
    } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t x = smh_wm_pairs_prep(w[q]);
        cnt = smh_popc_add(smh_wm_pair_step_bits(c2, x, 0, tab), cnt);
        cnt = smh_popc_add(smh_wm_pair_step_bits(c2, x, 1, tab), cnt);
    }
    return cnt;
    }
}

/* slow path of the pair kernel: per-column test against the exact m-symbol filter held in HBM */
SMH_LANE uint32_t smh_wm_pair_lane_slow(const uint8_t *text, uint64_t n, uint64_t a, const uint32_t *filter_g, int m,
                                        uint64_t *match_mask = nullptr)
{
    if (match_mask) *match_mask = 0;
    if (a >= n) return 0;
    uint64_t end = a + SMH_SEG;
    if (end > n) end = n;
    uint64_t e0 = a;
    if (e0 < (uint64_t)(m - 1)) e0 = (uint64_t)(m - 1);
    if (e0 >= end) return 0;
    const uint32_t mask = (1u << (2 * m)) - 1u;
    uint32_t code = 0, cnt = 0;
    for (uint64_t i = e0 - (uint64_t)(m - 1); i < e0; ++i) code = (code << 2) | (text[i] & 3u);
    for (uint64_t e = e0; e < end; ++e) {
        code = (code << 2) | (text[e] & 3u);
        const uint32_t key = code & mask;
        const uint32_t hit = (filter_g[key >> 5] >> (key & 31u)) & 1u;
        cnt += hit;
        if (match_mask && hit) *match_mask |= 1ull << (e - a);
    }
    return cnt;
}

template <bool PREFETCH, bool POS = false>
SMH_LANE uint32_t smh_wm_pair_thread(uint64_t gthread, const smh_chunk_sched &S, const uint8_t *text, uint64_t n, int m,
                                     const void *tab, const uint32_t *filter_g, const smh_pos_out *po = nullptr)
{
    if (n < (uint64_t)m) return 0;
    const uint64_t chunk_bytes = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (n + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    uint32_t cnt = 0;
    uint32_t cur[16], nxt[16], cur_edge[2], nxt_edge[2];
    uint64_t k = S.take(n_chunks);
    /* chunk 0 has no text in front of it and columns < m-1 without a window: slow path */
    auto is_fast = [&](uint64_t kk) { return kk >= 1 && kk < n_chunks && (kk + 1) * chunk_bytes <= n; };
    auto load = [&](uint64_t kk, uint32_t (&w)[16], uint32_t (&edge)[2]) {
        const uint64_t base = smh_uniform64(kk * chunk_bytes);
        const uint8_t *p = text + base + (uint64_t)lane * SMH_SEG;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const smh_u32x4 t = smh_load16(p + 16u * q);
            w[4 * q + 0] = t.v[0];
            w[4 * q + 1] = t.v[1];
            w[4 * q + 2] = t.v[2];
            w[4 * q + 3] = t.v[3];
        }
        /* the 8 bytes in front of the wave-chunk, same address in every lane (16-byte aligned load) */
        const smh_u32x4 t = smh_load16(text + base - 16u);
        edge[0] = t.v[2];
        edge[1] = t.v[3];
    };
    bool cur_fast = is_fast(k);
    if (cur_fast) load(k, cur, cur_edge);
    while (k < n_chunks) {
        const uint64_t kn = S.take(n_chunks);
        const bool nxt_fast = is_fast(kn);
        if (PREFETCH && nxt_fast) load(kn, nxt, nxt_edge);
        const uint64_t a = smh_uniform64(k * chunk_bytes) + (uint64_t)lane * SMH_SEG;
        if (cur_fast) {
            cnt += smh_wm_pair_lane_fast<POS>(text, a, cur, cur_edge, tab, po);
        } else if (POS) {
            uint64_t mm;
            smh_wm_pair_lane_slow(text, n, a, filter_g, m, &mm);
            cnt += smh_append_bits(mm, a, *po);
        } else {
            cnt += smh_wm_pair_lane_slow(text, n, a, filter_g, m);
        }
        if (nxt_fast) {
            if (PREFETCH) {
#pragma unroll
                for (int q = 0; q < 16; ++q) cur[q] = nxt[q];
                cur_edge[0] = nxt_edge[0];
                cur_edge[1] = nxt_edge[1];
            } else {
                load(kn, cur, cur_edge);
            }
        }
        cur_fast = nxt_fast;
        k = kn;
    }
    return cnt;
}

/* ---- in-register verify (STG 5 / 6; round 3).  The staged verify costs a wave ~3000 cycles per chunk that has a surviving
 * column whatever their number -- try-lock, 4 KiB copy, queue and window reads are LDS round trips behind fifteen other
 * waves' lookups, and prefetching the next chunk hides next to none of it (gpurun_out/r03_r/prefetch.log) -- which is
 * what 8000 DNA patterns of 16 symbols pay at 4.4 survivors per chunk (0.26 against 0.174 ms/GiB for the bare scan).
 * With FEW survivors the lane that found one can hash the window itself: the window lies in its own 16 text registers
 * and the previous lane's last 4 * HP (one DPP move each), at a per-lane dword offset sd that registers cannot be
 * indexed with -- so the MAXD + 1 dwords from sd on are brought into place by a barrel of conditional moves, one level
 * per bit of sd (39 v_cndmask for HP = 1, 59 for HP = 2; all lanes execute them, each for its own column).  No LDS, no
 * lock, no compaction; the bucket probe is software-pipelined as before, one pending column PER LANE.  A lane with a
 * second survivor in the same chunk (rare by the launcher's choice of this form) decides its first one at once. */

/* one level of the barrel: `mask` ? t : f per lane.  On the GPU an explicit v_cndmask_b32 on the ballot of the level's
 * condition: written as a C++ conditional over local arrays, LLVM turns "c ? x[i + 8] : x[i]" into a LOAD from a selected
 * address, i.e. puts the arrays into scratch memory (measured: the kernel 37 % slower even on sets without survivors) */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
typedef uint64_t smh_lanemask;
SMH_LANE smh_lanemask smh_lane_mask(bool b) { return __ballot(b); }
SMH_LANE uint32_t smh_sel(smh_lanemask mask, uint32_t t, uint32_t f)
{
    uint32_t d;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(d) : "v"(f), "v"(t), "s"(mask));
    return d;
}
#else
typedef bool smh_lanemask;
SMH_LANE smh_lanemask smh_lane_mask(bool b) { return b; }
SMH_LANE uint32_t smh_sel(smh_lanemask mask, uint32_t t, uint32_t f) { return mask ? t : f; }
#endif

/* hash of the m-byte window that ends at column c (0..63) of the lane's segment: w = the segment, prev = the 16 * HP
 * bytes in front of it */
template <int HP>
SMH_LANE uint32_t smh_regv_tag(const uint32_t (&w)[16], const uint32_t (&prev)[4 * HP], uint32_t c, int m)
{
    constexpr int NP = 4 * HP, NA = NP + 16, MAXD = NP + 1; /* MAXD == SMH_STAGE_MAXD(HP) */
    const uint32_t s0 = 16u * HP + c + 1u - (uint32_t)m; /* first byte of the window, counted from prev[0]; m - 1 <= 16 * HP */
    const uint32_t sd = s0 >> 2, sh = (s0 & 3u) * 8u;    /* sd <= NP + 15 */
    /* level 16: only the first NP entries can come from 16 further on (sd >= 16 leaves a shift of < NP behind it; what
     * lies beyond the segment's last dword is outside every window and never reaches the hash) */
    uint32_t x4[MAXD + 16];
    {
        const smh_lanemask b = smh_lane_mask((sd & 16u) != 0);
#pragma unroll
        for (int i = 0; i < MAXD + 16; ++i) {
            const uint32_t lo = i < NP ? prev[i < NP ? i : 0] : (i < NA ? w[i < NA && i >= NP ? i - NP : 0] : 0u);
            x4[i] = i < NP ? smh_sel(b, w[i < NP ? i + 16 - NP : 0], lo) : lo;
        }
    }
    uint32_t x3[MAXD + 8];
    {
        const smh_lanemask b = smh_lane_mask((sd & 8u) != 0);
#pragma unroll
        for (int i = 0; i < MAXD + 8; ++i) x3[i] = smh_sel(b, x4[i + 8], x4[i]);
    }
    uint32_t x2[MAXD + 4];
    {
        const smh_lanemask b = smh_lane_mask((sd & 4u) != 0);
#pragma unroll
        for (int i = 0; i < MAXD + 4; ++i) x2[i] = smh_sel(b, x3[i + 4], x3[i]);
    }
    uint32_t x1[MAXD + 2];
    {
        const smh_lanemask b = smh_lane_mask((sd & 2u) != 0);
#pragma unroll
        for (int i = 0; i < MAXD + 2; ++i) x1[i] = smh_sel(b, x2[i + 2], x2[i]);
    }
    uint32_t d[MAXD + 1];
    {
        const smh_lanemask b = smh_lane_mask((sd & 1u) != 0);
#pragma unroll
        for (int i = 0; i <= MAXD; ++i) d[i] = smh_sel(b, x1[i + 1], x1[i]);
    }
    return smh_wm_tag_dwords<MAXD>(d, sh, m);
}

/* the surviving columns `msk` (bit b = column a + b) of the lane's segment; all 64 lanes must call it */
template <int HP>
SMH_LANE void smh_wm_regv_columns(smh_wm_queue &Q, const uint8_t *text, uint64_t a, uint64_t msk, const uint32_t (&w)[16],
                                  const uint32_t (&halo)[4 * HP], const smh_wm_params &P)
{
    if (!SMH_WAVE_ANY(msk != 0)) return;
    uint32_t prev[4 * HP]; /* lane 0: the bytes in front of the wave-chunk */
#pragma unroll
    for (int q = 0; q < 4 * HP; ++q) prev[q] = smh_prev_lane_word(w[16 - 4 * HP + q], halo[q], text, a - 16u * HP + 4u * (uint32_t)q);
    do {
        const bool have = msk != 0;
        const uint32_t c = have ? (uint32_t)__builtin_ctzll(msk) : 0u;
        const uint32_t tag = smh_regv_tag<HP>(w, prev, c, P.m);
        /* the column this lane (or another) still holds from an earlier chunk -- its bucket arrived while this chunk was
         * scanned -- or from the round before (a second survivor in one lane: decided without the pipelining) */
        smh_wm_pend_finish_rv(Q, text, P);
        Q.events += have ? 1u : 0u;
        Q.pend_n = 64u;
        Q.pend_mine = have ? 1u : 0u;
        Q.pend_e = a + c;
        Q.pend_tag = tag;
        msk &= msk - 1u;
    } while (SMH_WAVE_ANY(msk != 0));
}

/* ------------------------------------------------------------------ gram filter (q-gram shift-or)
 * smh_internal.h "gram filter" describes the tables.  A lane owns the 64 END columns of its segment, keeps the
 * shift-or state S in one register (bit b CLEAR = "the last b+1 grams are in planes b .. 0 in order"; a column is a
 * candidate when bit 7 is clear after its step) and never looks back further than the q-1 symbols in front of the
 * segment (they come out of the previous lane's registers, as in the pair kernel): the state it would have
 * inherited from the columns before the segment is ASSUMED all-alive (low seven bits clear), and the first seven
 * candidate bits are corrected afterwards with the previous lane's final state (one DPP move) -- bit 6-t of that
 * state is exactly the term the assumption replaced in column t.  Lane 0 of a wave has no neighbour and keeps the
 * assumption: a few more columns reach the verify stage, which is exact, so the count does not change.
 * One column costs ONE v_lshl_or_b32: S = (S << 1) | G with G the table byte (zero-extended by ds_read_u8), so the
 * candidate bits just keep shifting up the register and 24 columns' flags are collected with one bit-reverse.  The
 * pair form does TWO columns with one v_lshl_or_b32: its 16-bit entry is (G_first << 1) | G_second, with G up to 15 bits
 * wide (J planes, candidate = bit J-1 clear; flags collected every 16 columns). */
#define SMH_GRAM_S0 0xFFFFFF80u /* no candidates behind, all seven inherited terms alive */

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE uint32_t smh_lds_u8(const void *, uint32_t byte_off)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>(byte_off);
}
SMH_LANE uint32_t smh_bitrev32(uint32_t v) { return __builtin_bitreverse32(v); }
SMH_LANE uint32_t smh_mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
#else
SMH_LANE uint32_t smh_lds_u8(const void *base, uint32_t byte_off) { return ((const uint8_t *)base)[byte_off]; }
SMH_LANE uint32_t smh_bitrev32(uint32_t v)
{
    uint32_t r = 0;
    for (int i = 0; i < 32; ++i) r |= ((v >> i) & 1u) << (31 - i);
    return r;
}
SMH_LANE uint32_t smh_mul24(uint32_t a, uint32_t b) { return (uint32_t)((uint64_t)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)); }
#endif

/* one column of the recurrence; G = the table byte (zero-extended) */
SMH_LANE uint32_t smh_gram_step(uint32_t S, uint32_t G) { return (S << 1) | G; }
/* two columns; E = (G of the first << 1) | G of the second */
SMH_LANE uint32_t smh_gram_step2(uint32_t S, uint32_t E) { return (S << 2) | E; }

/* candidate flags of the last `cols` (<= 24) columns, bit c SET = the c-th of them (oldest first) is a candidate */
SMH_LANE uint32_t smh_gram_flags(uint32_t S, int cols)
{
    /* after the group's last column, bit 7 + t of S is the (inverted) flag of the column t before it */
    return (smh_bitrev32(~S) >> (25 - cols)) & ((1u << cols) - 1u);
}

/* pair form with J planes: candidate flags of the last 16 columns (bit c SET = the c-th of them, oldest first) */
SMH_LANE uint32_t smh_gram_flags16(uint32_t S, int J)
{
    const uint32_t s = J >= 8 ? S >> (J - 8) : S << (8 - J); /* candidate bit J-1 -> bit 7 */
    return (smh_bitrev32(~s) >> 9) & 0xFFFFu;
}

/* SMH_GRAM_OCT2 (J planes): after the eighth lookup of a group of 16 columns the candidate bits of its lookups t = 0..7
 * stand at bits J-2+2(7-t) (END = the column after lookup t's pair) and J-1+2(7-t) (END = the pair's second column):
 * bit o SET of the result = the column at offset o from the group's SECOND column (its first one's flag came with the
 * lookup in front of the group) is a candidate */
SMH_LANE uint32_t smh_gram_flags16_oct2(uint32_t S, int J) { return (smh_bitrev32(~(S >> (J - 2))) >> 16) & 0xFFFFu; }

/* byte-gram key of column i of the segment: the three bytes that end there, as the low 24 bits of a dword.
 * `pre` holds the four bytes in front of the segment. */
template <int I>
SMH_LANE uint32_t smh_gram_key(const uint32_t (&w)[16], uint32_t pre)
{
    if constexpr (I == 0) return smh_alignbyte(w[0], pre, 2u);
    else if constexpr (I == 1) return smh_alignbyte(w[0], pre, 3u);
    else {
        constexpr int first = I - 2, d = first >> 2, r = first & 3;
        if constexpr (r == 0) return w[d];
        else if constexpr (d == 15) return smh_alignbyte(0u, w[15], (uint32_t)r); /* the byte past the segment is not part of the gram */
        else return smh_alignbyte(w[d + 1], w[d], (uint32_t)r);
    }
}

/* the FOUR bytes that end at column i (KIND 11): the dword at byte offset i - 3 of the segment */
template <int I>
SMH_LANE uint32_t smh_gram_key4(const uint32_t (&w)[16], uint32_t pre)
{
    if constexpr (I < 3) return smh_alignbyte(w[0], pre, (uint32_t)(I + 1));
    else {
        constexpr int first = I - 3, d = first >> 2, r = first & 3;
        if constexpr (r == 0) return w[d];
        else return smh_alignbyte(w[d + 1], w[d], (uint32_t)r); /* d + 1 <= 15: the gram ENDS inside the segment */
    }
}
/* its product: low three bytes x SMH_GRAM_MUL + fourth byte x SMH_GRAM_MUL4 (== SMH_GRAM_PROD4): an SDWA multiply on the top byte and a
 * multiply-add -- one vector instruction more than the three-byte gram's */
SMH_LANE uint32_t smh_gram_prod4(uint32_t key32)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    uint32_t t;
    const uint32_t mul4 = 0x9E3779u; /* SMH_GRAM_MUL4 */
    asm("v_mul_u32_u24_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(t) : "v"(key32), "v"(mul4));
    return __umul24(key32, SMH_GRAM_MUL_DEV) + t;
#else
    return (uint32_t)((uint64_t)(key32 & 0xFFFFFFu) * SMH_GRAM_MUL_DEV) + (uint32_t)((uint64_t)(key32 >> 24) * 0x9E3779u);
#endif
}

/* G of a byte-gram column from its key (the three bytes that end there, in the low 24 bits).
 *   KIND 2  one plane per offset: the table byte at the top 17 bits of key * SMH_GRAM_MUL
 *   KIND 6  (round 3, SMH_GRAM_FLAT) ONE set for the grams of all offsets, a 2^20-bit array: byte address = the same 17
 *           bits, bit = the three below them; the array holds the set INVERTED, so a sign-extending 1-bit field extract
 *           yields 0 (in the set) or all ones (not), masked to the J plane bits `gmask` -- every plane tests the same set */
/* kernel KINDs of the byte-gram forms: 2 = hashed planes, 6 / 7 = flat set with one / two bits per gram; round 6: 8 / 9 / 10 = the same three
 * in the 143.9 KiB table (SMH_GRAM_BIG_BYTES) */
constexpr bool smh_kind_flat(int k) { return k == 6 || k == 7 || k == 9 || k == 10 || k == 11; }
constexpr bool smh_kind_flat_k2(int k) { return k == 7 || k == 10; }
constexpr bool smh_kind_big(int k) { return k == 8 || k == 9 || k == 10 || k == 11; }
constexpr bool smh_kind_wide(int k) { return k == 11; } /* late round 6: FOUR-byte grams (smh_internal.h SMH_GRAM_FLAT4_BIG) */
/* the instance of smh_gram_byte_G that looks a column up for kernel KIND k: 2 / 8 the planes, 6 / 9 the flat set */
constexpr int smh_kind_lookup(int k) { return k == 11 ? 11 : smh_kind_flat(k) ? (smh_kind_big(k) ? 9 : 6) : (k == 8 ? 8 : 2); }
/* KIND 8 (round 6): KIND 2 with a table of SMH_GRAM_BIG_BYTES bytes -- index = the product's low 24 bits scaled to the table's
 * DWORDS (v_mul_hi_u32_u24 yields 16 bits), the product's top two bits as the byte: == SMH_GRAM_BIG_INDEX in smh_internal.h */
SMH_LANE uint32_t smh_gram_big_index(uint32_t prod)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    uint32_t dw;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(dw) : "v"(prod), "s"((SMH_GRAM_BIG_BYTES_DEV / 4u) << 8));
    return __builtin_amdgcn_alignbit(dw, prod, 30u); /* (dw << 2) | (prod >> 30) */
#else
    return ((uint32_t)(((uint64_t)(prod & 0xFFFFFFu) * (uint64_t)((SMH_GRAM_BIG_BYTES_DEV / 4u) << 8)) >> 32) << 2) | (prod >> 30);
#endif
}
/* the flat set in the big table (KIND 9 / 10) is read a DWORD at a time: byte address = the dword index << 2, bit = the product's low
 * five bits -- v_lshrrev takes a shift's low five bits by itself, so the form costs what the 128 KiB one costs: mul, mul_hi, shift,
 * [ds_read_b32], shift, alignbit.  Second bit (KIND 10): the product's bits 24..28 (smh_flat_big_second).  The dword index is made of the
 * product's bits 9..23 (v_mul_hi_u32_u24 reads the low 24), so the "next five" (bits 5..9, the first build of round 6) shared
 * bit 9 with it: within a dword every second bit fell into one half, a random gram passed the second test at 0.55 where the
 * set is 0.37 full, and two bits filtered no better than one (m = 5: 1.05 % against 1.13 % of the columns).  Bits 27..31 pass 0.52 %
 * (independent bits: 0.47 %) at three instructions for the second bit; bits 24..28 -- bit 24 is a weaker one -- 0.64 % at two, which
 * measured 1.3 % faster: kept. */
SMH_LANE uint32_t smh_gram_big_dword(uint32_t prod)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    uint32_t dw;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(dw) : "v"(prod), "s"((SMH_GRAM_BIG_BYTES_DEV / 4u) << 8));
    return dw << 2;
#else
    return (uint32_t)(((uint64_t)(prod & 0xFFFFFFu) * (uint64_t)((SMH_GRAM_BIG_BYTES_DEV / 4u) << 8)) >> 32) << 2;
#endif
}
/* KIND 10's second bit: bit (product bits 24..28) of the dword -- the shift takes the index out of the product's top BYTE by itself
 * (SDWA byte select; the shift uses an amount's low five bits), so the second bit costs a shift and an or */
SMH_LANE uint32_t smh_flat_big_second(uint32_t word, uint32_t prod)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    uint32_t t;
    asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(t) : "v"(prod), "v"(word));
    return t;
#else
    return word >> ((prod >> 24) & 31u);
#endif
}
template <int KIND>
SMH_LANE uint32_t smh_gram_byte_G(uint32_t key, const void *tab, uint32_t gmask, bool k2 = false)
{
    if constexpr (KIND == 11) { /* four-byte grams, one bit (bounds-checked path): key = all four bytes */
        const uint32_t prod4 = smh_gram_prod4(key);
        return ((smh_lds_u32(tab, smh_gram_big_dword(prod4)) >> (prod4 & 31u)) & 1u) ? gmask : 0u;
    }
    const uint32_t prod = smh_mul24(key, SMH_GRAM_MUL_DEV);
    if constexpr (KIND == 9) { /* the flat set in the big table (bounds-checked path) */
        const uint32_t word = smh_lds_u32(tab, smh_gram_big_dword(prod));
        uint32_t out = word >> (prod & 31u);
        if (k2) out |= smh_flat_big_second(word, prod);
        return (out & 1u) ? gmask : 0u;
    }
    const uint32_t b = smh_lds_u8(tab, KIND == 8 ? smh_gram_big_index(prod) : prod >> 15);
    if constexpr (KIND == 6) { /* the bounds-checked path (the fast path: smh_flat_columns); k2: a gram has TWO bits in its byte */
        uint32_t out = b >> ((prod >> 12) & 7u); /* bit indices: product bits 12..14 and 9..11, under the 17 address bits */
        if (k2) out |= b >> ((prod >> 9) & 7u);
        return (out & 1u) ? gmask : 0u;
    } else {
        (void)k2;
        (void)gmask;
        return b;
    }
}

template <int KIND, int... Is>
SMH_LANE void smh_gram_byte_columns(const uint32_t (&w)[16], uint32_t pre, const void *tab, uint32_t gmask, uint32_t &T, uint32_t (&fl)[3],
                                    std::integer_sequence<int, Is...>)
{
    /* columns in order; flags are collected after columns 23, 47 and 63 */
    ((T = smh_gram_step(T, smh_gram_byte_G<KIND>(smh_gram_key<Is>(w, pre), tab, gmask)),
      (Is == 23 ? (void)(fl[0] = smh_gram_flags(T, 24)) : Is == 47 ? (void)(fl[1] = smh_gram_flags(T, 24))
                                                        : Is == 63 ? (void)(fl[2] = smh_gram_flags(T, 16)) : (void)0)),
     ...);
}

/* ---- flat byte grams (KIND 6), round 4: a column's lookup yields ONE bit -- "this 3-byte gram is in no pattern" -- and the
 * filter asks whether the last J columns' bits are all clear.  The plane-masked shift-or state of the other forms cost this
 * one 8.7 VALU and one quarter-rate multiply per column (the bit became a plane mask with v_bfe_i32 + v_and, and the compiler
 * turned the chain of shift-ors into a tree with v_mul_lo_u32): 12.7 issue slots per column, 76 % of the kernel.  Now the
 * bits are simply COLLECTED -- H = alignbit(byte >> index, H, 1): after 32 columns bit i is column i's -- which is
 * alignbyte, mul24, shift (address), shift (index), shift (bit), alignbit = 5.75 VALU per column, and the "last J all
 * clear" test runs once per segment on the 96 collected bits (the previous lane's last 32 in front) with three or four
 * shifted ORs. */
SMH_LANE uint32_t smh_flat_push(uint32_t H, uint32_t t)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    return __builtin_amdgcn_alignbit(t, H, 1u);
#else
    return (H >> 1) | ((t & 1u) << 31);
#endif
}
/* K2 (round 4, patterns of 5..7 bytes): a gram has TWO bits in its byte (indices = product bits 12..14 and 9..11) and is in
 * the set when both are clear in the inverted array -- a blocked Bloom filter with 8-bit blocks.  With J = 3 grams of 100 000
 * patterns one bit per gram fills 25 % of the 2^20 bits and passes 0.25^3 = 1.6 % of random columns (66 per 4 KiB); two bits
 * fill 44 % and pass (0.44^2)^3 = 0.7 %.  From six grams on the fuller array loses and the compile keeps one bit (wm_host.c). */
/* the bit index (product bits 12..14; K2's second one: bits 9..11) out of an asm v_bfe_u32: written as C++ the compiler
 * recomputes "bits 12..14 of a 24-bit product" as the top bits of a second, FULL 32-bit multiply (v_mul_lo_u32: quarter
 * rate) per column */
template <int OFF>
SMH_LANE uint32_t smh_flat_idx(uint32_t prod)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    uint32_t idx;
    if constexpr (OFF == 12) asm("v_bfe_u32 %0, %1, 12, 3" : "=v"(idx) : "v"(prod));
    else asm("v_bfe_u32 %0, %1, 9, 3" : "=v"(idx) : "v"(prod));
    return idx;
#else
    return (prod >> OFF) & 7u;
#endif
}
SMH_LANE uint32_t smh_flat_bit(uint32_t key, const void *tab, bool k2, bool big, bool wide = false)
{
    if (wide) { /* KIND 11: key = the four bytes, one bit, big table */
        const uint32_t prod4 = smh_gram_prod4(key);
        return smh_lds_u32(tab, smh_gram_big_dword(prod4)) >> (prod4 & 31u);
    }
    const uint32_t prod = smh_mul24(key, SMH_GRAM_MUL_DEV);
    if (big) {
        const uint32_t word = smh_lds_u32(tab, smh_gram_big_dword(prod));
        return k2 ? (word >> (prod & 31u)) | smh_flat_big_second(word, prod) : word >> (prod & 31u);
    }
    const uint32_t b = smh_lds_u8(tab, prod >> 15);
    return k2 ? (b >> smh_flat_idx<12>(prod)) | (b >> smh_flat_idx<9>(prod)) : b >> smh_flat_idx<12>(prod);
}
/* eight columns: the products, the eight lookups in flight together, then the bits (only the products and the bytes live
 * across the lookups: with the indices kept beside them the two-bit variant ran out of registers) */
template <int G, bool K2, bool BIG, bool WIDE = false>
SMH_LANE void smh_flat_group(const uint32_t (&w)[16], uint32_t pre, const void *tab, uint32_t &H)
{
    uint32_t prod[8], b[8];
    if constexpr (WIDE) { /* four-byte grams (KIND 11) */
        prod[0] = smh_gram_prod4(smh_gram_key4<8 * G + 0>(w, pre));
        prod[1] = smh_gram_prod4(smh_gram_key4<8 * G + 1>(w, pre));
        prod[2] = smh_gram_prod4(smh_gram_key4<8 * G + 2>(w, pre));
        prod[3] = smh_gram_prod4(smh_gram_key4<8 * G + 3>(w, pre));
        prod[4] = smh_gram_prod4(smh_gram_key4<8 * G + 4>(w, pre));
        prod[5] = smh_gram_prod4(smh_gram_key4<8 * G + 5>(w, pre));
        prod[6] = smh_gram_prod4(smh_gram_key4<8 * G + 6>(w, pre));
        prod[7] = smh_gram_prod4(smh_gram_key4<8 * G + 7>(w, pre));
    } else {
    prod[0] = smh_mul24(smh_gram_key<8 * G + 0>(w, pre), SMH_GRAM_MUL_DEV);
    prod[1] = smh_mul24(smh_gram_key<8 * G + 1>(w, pre), SMH_GRAM_MUL_DEV);
    prod[2] = smh_mul24(smh_gram_key<8 * G + 2>(w, pre), SMH_GRAM_MUL_DEV);
    prod[3] = smh_mul24(smh_gram_key<8 * G + 3>(w, pre), SMH_GRAM_MUL_DEV);
    prod[4] = smh_mul24(smh_gram_key<8 * G + 4>(w, pre), SMH_GRAM_MUL_DEV);
    prod[5] = smh_mul24(smh_gram_key<8 * G + 5>(w, pre), SMH_GRAM_MUL_DEV);
    prod[6] = smh_mul24(smh_gram_key<8 * G + 6>(w, pre), SMH_GRAM_MUL_DEV);
    prod[7] = smh_mul24(smh_gram_key<8 * G + 7>(w, pre), SMH_GRAM_MUL_DEV);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = BIG ? smh_lds_u32(tab, smh_gram_big_dword(prod[j])) : smh_lds_u8(tab, prod[j] >> 15);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t t;
        if constexpr (BIG) { /* a dword of the set: the bit index is the product's low five bits, the shift takes them by itself */
            t = b[j] >> (prod[j] & 31u);
            if constexpr (K2) t |= smh_flat_big_second(b[j], prod[j]);
        } else {
            t = b[j] >> smh_flat_idx<12>(prod[j]);
            if constexpr (K2) t |= b[j] >> smh_flat_idx<9>(prod[j]);
        }
        H = smh_flat_push(H, t);
    }
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    /* the groups do not depend on each other except through H, and left to itself the scheduler starts all eight at once:
     * 128 VGPRs and spills to scratch.  One group's lookups in flight are what the LDS needs. */
    __builtin_amdgcn_sched_barrier(0);
#endif
}
template <bool K2, bool BIG, bool WIDE = false>
SMH_LANE void smh_flat_columns(const uint32_t (&w)[16], uint32_t pre, const void *tab, uint32_t &H0, uint32_t &H1)
{
    smh_flat_group<0, K2, BIG, WIDE>(w, pre, tab, H0);
    smh_flat_group<1, K2, BIG, WIDE>(w, pre, tab, H0);
    smh_flat_group<2, K2, BIG, WIDE>(w, pre, tab, H0);
    smh_flat_group<3, K2, BIG, WIDE>(w, pre, tab, H0);
    smh_flat_group<4, K2, BIG, WIDE>(w, pre, tab, H1);
    smh_flat_group<5, K2, BIG, WIDE>(w, pre, tab, H1);
    smh_flat_group<6, K2, BIG, WIDE>(w, pre, tab, H1);
    smh_flat_group<7, K2, BIG, WIDE>(w, pre, tab, H1);
}
/* (hi:lo) << k for 0 < k < 32: bit i of the result = bit i - k of the 64-bit sequence whose upper word is hi */
SMH_LANE uint32_t smh_shl_across(uint32_t hi, uint32_t lo, uint32_t k) { return (hi << k) | (lo >> (32u - k)); }
/* columns of the segment (bit c of the result) whose own bit and the J - 1 before it are all clear; Z0 = the 32 columns in
 * front of the segment, Z1 / Z2 = its columns 0..31 / 32..63 (bit SET = not in the set); 1 <= J <= 32 */
SMH_LANE uint64_t smh_flat_candidates(uint32_t Z0, uint32_t Z1, uint32_t Z2, uint32_t J)
{
    uint32_t width = 1;
    while (2u * width <= J) { /* wave-uniform: J is a launch parameter */
        const uint32_t n2 = Z2 | smh_shl_across(Z2, Z1, width), n1 = Z1 | smh_shl_across(Z1, Z0, width);
        Z0 |= Z0 << width;
        Z1 = n1;
        Z2 = n2;
        width *= 2u;
    }
    if (width < J) {
        const uint32_t k = J - width;
        const uint32_t n2 = Z2 | smh_shl_across(Z2, Z1, k), n1 = Z1 | smh_shl_across(Z1, Z0, k);
        Z1 = n1;
        Z2 = n2;
    }
    return ~(((uint64_t)Z2 << 32) | Z1);
}
/* the 32 bits in front of the segment at a, true values (the emulator; the GPU takes the previous lane's H1) */
SMH_LANE uint32_t smh_flat_history_before(const uint8_t *text, uint64_t a, const void *tab, bool k2, bool big, bool wide = false)
{
    uint32_t H = 0;
    for (uint64_t x = a >= 32 ? a - 32 : 0; x < a; ++x) {
        uint32_t t = 0; /* a column without a whole gram in front of it cannot be ruled out */
        if (wide) {
            if (x >= 3) t = smh_flat_bit((uint32_t)text[x - 3] | ((uint32_t)text[x - 2] << 8) | ((uint32_t)text[x - 1] << 16) | ((uint32_t)text[x] << 24), tab, false, true, true);
        } else
        if (x >= 2) t = smh_flat_bit((uint32_t)text[x - 2] | ((uint32_t)text[x - 1] << 8) | ((uint32_t)text[x] << 16), tab, k2, big);
        H = smh_flat_push(H, t);
    }
    return H; /* (a < 32 never reaches the fast path: chunk 0 is bounds-checked) */
}

/* state the lane would have inherited from the columns in front of its segment (low 7 bits, 0 = alive), true
 * value: the CPU emulation and the bounds-checked path compute it by running the recurrence over those columns.
 * `tab` = the LDS image; `g7` = the pair form's per-gram bytes in HBM (smh_wm_params::gram_g7). */
template <int KIND>
SMH_LANE uint32_t smh_gram_state_before(const uint8_t *text, uint64_t a, const void *tab, const uint8_t *g7, uint32_t gmask = 0xFFu, bool k2 = false)
{
    uint32_t S = 0u;
    if (KIND == 3) {
        if (a < 15) return S;
        for (uint64_t x = a - 7; x < a; ++x) {
            uint32_t code = 0;
            for (int i = 7; i >= 0; --i) code = (code << 2) | (text[x - (uint64_t)i] & 3u);
            S = smh_gram_step(S, smh_lds_u8(tab, code));
        }
    } else if (KIND == 1) {
        /* J - 1 columns of history (J = planes, handed over in place of a table pointer's companion: see callers) */
        return 0u; /* the pair form has its own routine: smh_gram1_state_before */
    } else {
        if (a < 11) return S;
        for (uint64_t x = a - 7; x < a; ++x) {
            const uint32_t key = smh_kind_wide(KIND) ? (uint32_t)text[x - 3] | ((uint32_t)text[x - 2] << 8) | ((uint32_t)text[x - 1] << 16) | ((uint32_t)text[x] << 24)
                                                     : (uint32_t)text[x - 2] | ((uint32_t)text[x - 1] << 8) | ((uint32_t)text[x] << 16);
            S = smh_gram_step(S, smh_gram_byte_G<smh_kind_lookup(KIND)>(key, tab, gmask, k2));
        }
    }
    return S & 0x7Fu;
}

/* pair form (KIND 1): the low J-1 state bits a lane would have inherited, true value (0 = alive) */
SMH_LANE uint32_t smh_gram1_state_before(const uint8_t *text, uint64_t a, const uint8_t *g7, int J)
{
    uint32_t S = 0u;
    if (a < (uint64_t)(J - 1) + 6u) return S; /* columns without seven symbols in front of them: keep the assumption (superset) */
    for (uint64_t x = a - (uint64_t)(J - 1); x < a; ++x) {
        uint32_t code = 0;
        for (int i = 6; i >= 0; --i) code = (code << 2) | (text[x - (uint64_t)i] & 3u);
        uint16_t g;
        memcpy(&g, g7 + 2u * code, 2);
        S = smh_gram_step(S, g);
    }
    return S & ((1u << (J - 1)) - 1u);
}

/* SMH_GRAM_OCT2 (KIND 5): the low J-1 state bits a lane inherits, true value (0 = alive): the lookups at the odd columns in
 * front of a (a is a multiple of 64), oldest first */
SMH_LANE uint32_t smh_gram5_state_before(const uint8_t *text, uint64_t a, const void *tab, int J)
{
    const int NL = J / 2 + 1; /* lookups that reach the low J-1 bits */
    if (a < (uint64_t)(2 * NL + 6)) return 0u; /* lookups without eight symbols in front of them: keep the assumption (superset) */
    uint32_t S = 0u;
    for (int t = NL - 1; t >= 0; --t) {
        const uint64_t c = a - 1u - 2u * (uint64_t)t;
        uint32_t code = 0;
        for (int i = 7; i >= 0; --i) code = (code << 2) | (text[c - (uint64_t)i] & 3u);
        S = smh_gram_step2(S, smh_lds_u16(tab, 2u * code));
    }
    return S & ((1u << (J - 1)) - 1u);
}

/* ---- grouped pairs (KIND 4, mixed-length sets; smh_internal.h SMH_GRAM_PAIR2): two shift-or states per lane.
 * State A as in the pair form (candidate = bit 7 clear); state B has jb planes, candidate = bit jb-1 clear. */
SMH_LANE void smh_gram2_state_before(const uint8_t *text, uint64_t a, const uint8_t *gx, uint32_t &SA, uint32_t &SB)
{
    SA = 0u;
    SB = 0u;
    if (a < 14) return; /* columns without seven symbols in front of them: keep the assumption (superset) */
    for (uint64_t x = a - 7; x < a; ++x) {
        uint32_t code = 0;
        for (int i = 6; i >= 0; --i) code = (code << 2) | (text[x - (uint64_t)i] & 3u);
        uint16_t g;
        memcpy(&g, gx + 2u * code, 2);
        SA = smh_gram_step(SA, g & 0xFFu);
        SB = smh_gram_step(SB, g >> 8);
    }
}

/* bounds-checked path of the grouped form (the text's first and last chunks).  Wave-uniform -- all 64 lanes walk their 64
 * columns in step, a lane past the text's end with nothing to do -- so that a candidate column is decided by the suffix
 * index like the fast path's (wave-wide chain loop and positions append) instead of class by class: with 25 classes a
 * candidate cost this path 25 window hashes, buckets and compares one after the other, ~50 us, and the wave that drew
 * chunk 0 of the headline's mixed set (one candidate in every four steps somewhere among its lanes) ran 0.4 ms behind all
 * the others -- the whole launch waited for it. */
SMH_LANE uint32_t smh_wm_gram2_lane_slow(const uint8_t *text, uint64_t n, uint64_t a, const smh_wm_params &P, const smh_pos_out *po)
{
    if (!SMH_WAVE_ANY(a < n)) return 0;
    uint64_t end = a + SMH_SEG;
    if (end > n) end = n;
    uint32_t SA = 0u, SB = 0u, cnt = 0;
    if (a < n) smh_gram2_state_before(text, a, P.gram_g7, SA, SB);
    for (uint32_t i = 0; i < SMH_SEG; ++i) {
        const uint64_t e = a + i;
        bool cand = false;
        if (e < end) {
            uint32_t GA = 0u, GB = 0u; /* a column without a whole gram in front of it cannot be ruled out */
            if (e + 1 >= 7u) {
                uint32_t code = 0;
                for (int k = 6; k >= 0; --k) code = (code << 2) | (text[e - (uint64_t)k] & 3u);
                uint16_t g;
                memcpy(&g, P.gram_g7 + 2u * code, 2);
                GA = g & 0xFFu;
                GB = g >> 8;
            }
            SA = smh_gram_step(SA, GA);
            SB = smh_gram_step(SB, GB);
            cand = (!((SA >> 7) & 1u) || (P.gram_jb && !((SB >> (P.gram_jb - 1)) & 1u))) && e + 1 >= (uint64_t)P.m;
        }
        if (!SMH_WAVE_ANY(cand)) continue;
        if (P.sfx_slot && n >= 64u && !SMH_WAVE_ANY(cand && e < 31u)) {
            cnt += smh_wm_verify_sfx(text, e, cand, P, po);
            continue;
        }
        for (int c = 0; c < P.n_classes; ++c) { /* a column in the text's first bytes, or no index: class by class */
            const uint32_t hit = cand ? smh_wm_verify_class(text, e, P, c) : 0u;
            cnt += hit;
            if (po) smh_append_bits(hit, e, *po);
        }
    }
    return cnt;
}

/* the shift-or state the wave-chunk inherits, worked out from the halo registers (the last ND dwords of the HD in front of
 * the chunk: the same in every lane, so the lookups are broadcasts): the pair forms' rolling code over the halo's symbols,
 * one smh_gram_step2 per lookup from the fourth pair on (the first three only fill the eight-symbol code) */
template <int HD, int ND>
SMH_LANE uint32_t smh_gram_halo_state(const uint32_t (&halo)[HD], const void *tab)
{
    static_assert(ND <= HD, "the halo kept in registers covers the lookups that reach the inherited state");
    uint32_t c2 = 0, Th = 0;
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const uint32_t hw = halo[HD - ND + d], x = (hw << 10) | hw;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            c2 = ((c2 << 4) & 0x1FFFEu) | smh_bfe(x, k == 0 ? 7 : 23, 5);
            if (2 * d + k >= 3) Th = smh_gram_step2(Th, smh_lds_u16(tab, c2));
        }
    }
    return Th;
}
/* is this the lane that owns a wave-chunk's first segment?  The GPU knows its lane id; the emulator, a lane at a time, sees it
 * in the segment's offset */
SMH_LANE bool smh_is_lane0(uint64_t a)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    (void)a;
    return (threadIdx.x & 63u) == 0;
#else
    return ((a >> 6) & 63u) == 0;
#endif
}
/* "does lane 0 of the wave see `cond`?"  wave-uniform on the GPU; the emulator answers for the lane it is running (a lane
 * other than lane 0 has no use for the answer: it keeps its own inherited state) */
SMH_LANE bool smh_lane0_any(bool cond, uint64_t a)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    (void)a;
    return __builtin_amdgcn_readfirstlane((int)cond) != 0;
#else
    return smh_is_lane0(a) && cond;
#endif
}

/* fast path: the 64 END columns of the segment at a (a >= 4096: not the text's first chunk; a + 64 <= n).
 * `edge` = the 8 bytes in front of the wave-chunk (wave-uniform).  Returns nothing: candidates go to the queue. */
template <int KIND, bool POS, int STG = 0, bool QD = true>
SMH_LANE void smh_wm_gram_lane_fast(const uint8_t *text, uint64_t a, const uint32_t (&w)[16], const uint32_t (&halo)[4 * smh_stg_hp(STG)],
                                    const void *tab, const smh_wm_params &P, smh_wm_queue &Q)
{
    /* halo = the 16 * HP bytes in front of the wave-chunk (wave-uniform); its last two dwords prime lane 0 */
    constexpr int HP = smh_stg_hp(STG), HD = 4 * HP;
    /* pair form: J planes (2..15), the low J-1 bits assumed alive */
    [[maybe_unused]] const uint32_t jw = KIND == 1 || KIND == 5 ? (uint32_t)P.gram_planes - 1u : 7u, jmask = (1u << jw) - 1u;
    [[maybe_unused]] uint32_t fl16[4] = {0, 0, 0, 0};
    /* OCT2: the low J-2 bits assumed alive; bits J-2 and J-1 of the inherited state are candidate bits of columns that
     * are decided elsewhere (the first column's: below, from the inherited state itself) */
    uint32_t T = KIND == 1 ? ~jmask : KIND == 5 ? ~(jmask >> 1) : SMH_GRAM_S0, fl[3] = {0, 0, 0};
    /* grouped pairs: the short group's state (jb planes: the low jb-1 bits assumed alive), its flags, and the shift
     * that brings its candidate bit (jb-1) to bit 7 */
    [[maybe_unused]] const uint32_t bup = KIND == 4 && P.gram_jb ? 8u - (uint32_t)P.gram_jb : 0u;
    [[maybe_unused]] uint32_t TB = KIND == 4 && P.gram_jb ? ~((1u << (P.gram_jb - 1)) - 1u) : ~0u, flb[3] = {0, 0, 0};
    const uint32_t pre0 = smh_prev_lane_word(w[14], halo[HD - 2], text, a - 8u);
    const uint32_t pre1 = smh_prev_lane_word(w[15], halo[HD - 1], text, a - 4u);
    if constexpr (KIND == 1) {
        /* rolling code * 2 (16-bit entries) of the last eight symbols; primed with the eight in front.  Per PAIR
         * of columns: v_bfe + v_lshl_or (code), v_and (address), ds_read_u16, v_lshl_or (both columns' steps). */
        uint32_t code2 = 0;
        {
            const uint32_t x0 = (pre0 << 10) | pre0, x1 = (pre1 << 10) | pre1;
            code2 = (code2 << 4) | smh_bfe(x0, 7, 5);
            code2 = (code2 << 4) | smh_bfe(x0, 23, 5);
            code2 = (code2 << 4) | smh_bfe(x1, 7, 5);
            code2 = (code2 << 4) | smh_bfe(x1, 23, 5);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t x = (w[q] << 10) | w[q]; /* pair codes * 2 at bits 7..11 and 23..27 (ac_lane.h smh_fmt_s2) */
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                /* the code is kept masked, so it IS the byte address: v_bfe, v_lshlrev, v_and_or per lookup */
                code2 = ((code2 << 4) & 0x1FFFEu) | smh_bfe(x, k == 0 ? 7 : 23, 5);
                T = smh_gram_step2(T, smh_lds_u16(tab, code2)); /* columns a + 4q + 2k and + 1 */
            }
            if ((q & 3) == 3) fl16[q >> 2] = smh_gram_flags16(T, P.gram_planes); /* J planes: flags every 16 columns */
        }
    } else if constexpr (KIND == 5) {
        /* 8-symbol grams, a lookup per two columns: the pair form's rolling code and step; the entry carries the planes
         * of ALL offsets and the state chains every second one (smh_internal.h SMH_GRAM_OCT2) */
        uint32_t code2 = 0;
        {
            const uint32_t x0 = (pre0 << 10) | pre0, x1 = (pre1 << 10) | pre1;
            code2 = (code2 << 4) | smh_bfe(x0, 7, 5);
            code2 = (code2 << 4) | smh_bfe(x0, 23, 5);
            code2 = (code2 << 4) | smh_bfe(x1, 7, 5);
            code2 = (code2 << 4) | smh_bfe(x1, 23, 5);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t x = (w[q] << 10) | w[q];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                code2 = ((code2 << 4) & 0x1FFFEu) | smh_bfe(x, k == 0 ? 7 : 23, 5);
                T = smh_gram_step2(T, smh_lds_u16(tab, code2)); /* decides END columns a + 4q + 2k + 1 and + 2 */
            }
            if ((q & 3) == 3) fl16[q >> 2] = smh_gram_flags16_oct2(T, P.gram_planes);
        }
    } else if constexpr (KIND == 4) {
        /* grouped pairs: the pair form's lookup, two states (A above B in the entry) */
        const uint32_t bsh = P.gram_jb ? (uint32_t)P.gram_jb + 1u : 0u, bmask = (1u << bsh) - 1u;
        uint32_t code2 = 0;
        {
            const uint32_t x0 = (pre0 << 10) | pre0, x1 = (pre1 << 10) | pre1;
            code2 = (code2 << 4) | smh_bfe(x0, 7, 5);
            code2 = (code2 << 4) | smh_bfe(x0, 23, 5);
            code2 = (code2 << 4) | smh_bfe(x1, 7, 5);
            code2 = (code2 << 4) | smh_bfe(x1, 23, 5);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t x = (w[q] << 10) | w[q];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                code2 = ((code2 << 4) & 0x1FFFEu) | smh_bfe(x, k == 0 ? 7 : 23, 5);
                const uint32_t e = smh_lds_u16(tab, code2);
                T = smh_gram_step2(T, e >> bsh);
                TB = smh_gram_step2(TB, e & bmask);
            }
            if (q == 5) { fl[0] = smh_gram_flags(T, 24); flb[0] = smh_gram_flags(TB << bup, 24); }
            if (q == 11) { fl[1] = smh_gram_flags(T, 24); flb[1] = smh_gram_flags(TB << bup, 24); }
            if (q == 15) { fl[2] = smh_gram_flags(T, 16); flb[2] = smh_gram_flags(TB << bup, 16); }
        }
    } else if constexpr (KIND == 3) {
        /* 8-symbol grams, one lookup per column: the rolling code takes a pair of symbols per update; the column
         * of the pair's first symbol is indexed by the code without its newest symbol */
        uint32_t code = 0;
        {
            const uint32_t x0 = (pre0 << 10) | pre0, x1 = (pre1 << 10) | pre1;
            code = (code << 4) | smh_bfe(x0, 8, 4);
            code = (code << 4) | smh_bfe(x0, 24, 4);
            code = (code << 4) | smh_bfe(x1, 8, 4);
            code = (code << 4) | smh_bfe(x1, 24, 4);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t x = (w[q] << 10) | w[q];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                code = (code << 4) | smh_bfe(x, k == 0 ? 8 : 24, 4);
                const uint32_t e0 = smh_lds_u8(tab, smh_bfe(code, 2, 16)), e1 = smh_lds_u8(tab, code & 0xFFFFu);
                T = smh_gram_step(T, e0);
                T = smh_gram_step(T, e1);
            }
            if (q == 5) fl[0] = smh_gram_flags(T, 24);
            if (q == 11) fl[1] = smh_gram_flags(T, 24);
            if (q == 15) fl[2] = smh_gram_flags(T, 16);
        }
    } else if constexpr (smh_kind_flat(KIND)) {
        /* flat byte grams: the columns' bits are collected, the J-in-a-row test runs once on all of them (above).  KIND 7 = the
         * same with two bits per gram -- a kernel instance of its own: as a wave-uniform branch around two copies of the loop
         * the compiler hoisted all 64 columns' products in front of the branch (128 VGPRs and spills to scratch) */
        (void)pre0;
        uint32_t H0 = 0, H1 = 0, Hp;
        smh_flat_columns<smh_kind_flat_k2(KIND), smh_kind_big(KIND), smh_kind_wide(KIND)>(w, pre1, tab, H0, H1);
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
        /* the 32 columns in front of the segment: the previous lane's second half; lane 0 of a wave has no neighbour and
         * assumes "all in the set" (a few more columns reach the verify stage, which is exact) */
        Hp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)H1, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
#else
        Hp = smh_flat_history_before(text, a, tab, smh_kind_flat_k2(KIND), smh_kind_big(KIND), smh_kind_wide(KIND));
#endif
        const uint64_t msk6 = smh_flat_candidates(Hp, H0, H1, (uint32_t)P.gram_planes);
        if constexpr (smh_stg_regv(STG)) {
            smh_wm_regv_columns<HP>(Q, text, a, msk6, w, halo, P);
        } else if constexpr (smh_stg_l2(STG)) {
            smh_wm_l2_columns<smh_stg_l2_maxd(STG)>(Q, text, smh_uniform64(a & ~(uint64_t)4095), a, msk6, P);
        } else if constexpr (STG > 0) {
            smh_wm_stage_columns<STG, QD, true>(Q, text, smh_uniform64(a & ~(uint64_t)4095), a, msk6, w, halo, P);
        } else {
            uint64_t mq = msk6;
            while (SMH_WAVE_ANY(mq != 0)) {
                if (Q.count + 64u > SMH_WM_QCAP) smh_wm_drain(Q, text, P);
                const bool have = mq != 0;
                const uint32_t b = have ? (uint32_t)__builtin_ctzll(mq) : 0u;
                smh_wm_emit(Q, text, P, have, a + b);
                mq &= mq - 1u;
            }
        }
        return;
    } else {
        (void)pre0;
        smh_gram_byte_columns<KIND>(w, pre1, tab, 0xFFu & ~((1u << (8 - P.gram_planes)) - 1u), T, fl, std::make_integer_sequence<int, 64>{});
    }
    /* correct the first seven columns with the state the previous lane ended in */
    uint32_t prevT;
    [[maybe_unused]] uint32_t prevB = 0;
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    prevT = (uint32_t)__builtin_amdgcn_update_dpp((int)SMH_GRAM_S0, (int)T, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
    if constexpr (KIND == 4) prevB = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)TB, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
#else
    if constexpr (KIND == 4) smh_gram2_state_before(text, a, P.gram_g7, prevT, prevB);
    else if constexpr (KIND == 1) prevT = smh_gram1_state_before(text, a, P.gram_g7, P.gram_planes);
    else if constexpr (KIND == 5) prevT = smh_gram5_state_before(text, a, tab, P.gram_planes);
    else prevT = smh_gram_state_before<KIND>(text, a, tab, P.gram_g7, 0xFFu & ~((1u << (8 - P.gram_planes)) - 1u));
#endif
    uint64_t msk;
    if constexpr (KIND == 5) {
        /* Column q <= J-2 of the segment is a candidate only if bit J-2-q of the inherited state is alive too (the chain of
         * every second offset reaches back into the previous lane's lookups); column 0 has nothing but that bit.  Lane 0 of
         * a wave has no neighbour: the state the chunk inherits is worked out from the halo -- the last NP lookups in front
         * of the chunk, the same in every lane, so the reads are broadcasts -- in EVERY chunk (with the assumption alone
         * column 0 of every chunk would reach the verify stage). */
        uint32_t pv = prevT;
        {
            constexpr int NP = HP == 1 ? 5 : 8, ND = (2 * NP + 6 + 3) / 4; /* lookups (>= J / 2), dwords of halo that hold their symbols */
            const uint32_t Th = smh_gram_halo_state<HD, ND>(halo, tab);
            if (smh_is_lane0(a)) pv = Th;
        }
        const uint32_t fixj = smh_bitrev32(~pv & jmask) >> (32u - jw); /* bit q SET = bit J-2-q of the inherited state alive; J >= 3 */
        /* fl16[k] bit o = column 16 k + 1 + o; the flag of column 64 belongs to the next lane */
        const uint64_t later = (uint64_t)(fl16[0] | (fl16[1] << 16)) | ((uint64_t)(fl16[2] | (fl16[3] << 16)) << 32);
        msk = ((later << 1) | 1u) & ((uint64_t)(fixj | ~jmask) | 0xFFFFFFFF00000000ull);
    } else if constexpr (KIND == 1) {
        /* bit t (t < J-1) SET = bit J-2-t of the inherited state alive; lane 0's default (all of SMH_GRAM_S0's low
         * seven bits clear) is the assumption for J <= 8 and merely a subset of it above: its bits 7.. read as dead,
         * which would lose candidates, so lane 0 is given the assumption explicitly */
        uint32_t pv = prevT;
        if (smh_is_lane0(a)) pv = 0u;
        if constexpr (STG > 0) {
            /* ... which lets column c of lane 0 through on c + 1 planes only: 0.06 surviving columns per wave-chunk at 1000
             * patterns, 0.6 at 8000 (one chunk in two staged for nothing: 8000 patterns of 32 symbols scanned at 0.226
             * ms/GiB with 0.2 real survivors per chunk).  So when lane 0 has such a flag, the state the chunk inherits is
             * computed from the halo -- the last NP pairs of columns in front of the chunk decide its low J-1 bits; same
             * values in every lane, the lookups are broadcasts -- and lane 0 is corrected like the others.  (The CPU
             * emulation, a lane at a time, takes the same decision from lane 0's own flags and runs the same arithmetic.) */
            if (P.gram_jb >= 0 /* the launcher's choice: wm_kernels.inc launch_gram_stg */ && smh_lane0_any((fl16[0] & jmask) != 0, a)) {
                constexpr int NP = HP == 1 ? 5 : 7, ND = (2 * NP + 6) / 4; /* pairs of columns (>= J-1 columns), dwords of halo */
                const uint32_t Th = smh_gram_halo_state<HD, ND>(halo, tab);
                if (smh_is_lane0(a)) pv = Th;
            }
        }
        const uint32_t fixj = jw ? smh_bitrev32(~pv & jmask) >> (32u - jw) : 0u;
        msk = (uint64_t)((fl16[0] & (fixj | ~jmask)) | (fl16[1] << 16)) | ((uint64_t)(fl16[2] | (fl16[3] << 16)) << 32);
    } else {
    const uint32_t fix7 = smh_bitrev32(~prevT & 0x7Fu) >> 25; /* bit t SET = bit 6-t of the inherited state alive */
    msk = (uint64_t)((fl[0] & (fix7 | ~0x7Fu)) | (fl[1] << 24)) | ((uint64_t)(fl[1] >> 8) << 32) | ((uint64_t)fl[2] << 48);
    }
    if constexpr (KIND == 4) {
        if (P.gram_jb) {
            /* the short group's flags, its first jb-1 columns corrected the same way (lane 0 keeps the assumption) */
            const uint32_t wb = (uint32_t)P.gram_jb - 1u, wmask = (1u << wb) - 1u;
            const uint32_t fixb = wb ? smh_bitrev32(~prevB & wmask) >> (32u - wb) : 0u;
            msk |= (uint64_t)((flb[0] & (fixb | ~wmask)) | (flb[1] << 24)) | ((uint64_t)(flb[1] >> 8) << 32) | ((uint64_t)flb[2] << 48);
        }
    }
    if constexpr (smh_stg_regv(STG)) {
        /* in-register verify: few surviving columns, each hashed by its own lane out of the text registers */
        smh_wm_regv_columns<HP>(Q, text, a, msk, w, halo, P);
    } else if constexpr (smh_stg_l2(STG)) {
        /* windows from L2: requested now, hashed after the next chunk's scan, decided after the one after that */
        smh_wm_l2_columns<smh_stg_l2_maxd(STG)>(Q, text, smh_uniform64(a & ~(uint64_t)4095), a, msk, P);
    } else if constexpr (STG > 0) {
        /* staged verify: chunks with many surviving columns hash their windows from an LDS copy of the chunk */
        smh_wm_stage_columns<STG, QD, true>(Q, text, smh_uniform64(a & ~(uint64_t)4095), a, msk, w, halo, P);
    } else {
    while (SMH_WAVE_ANY(msk != 0)) {
        if (Q.count + 64u > SMH_WM_QCAP) smh_wm_drain(Q, text, P);
        const bool have = msk != 0;
        const uint32_t b = have ? (uint32_t)__builtin_ctzll(msk) : 0u;
        smh_wm_emit(Q, text, P, have, a + b);
        msk &= msk - 1u;
    }
    }
}

/* bounds-checked path for the text's first and last chunks: the same recurrence column by column from memory,
 * with the true inherited state; candidates are verified on the spot */
template <int KIND>
SMH_LANE uint32_t smh_wm_gram_lane_slow(const uint8_t *text, uint64_t n, uint64_t a, const void *tab, const smh_wm_params &P,
                                        uint64_t *match_mask = nullptr)
{
    if (match_mask) *match_mask = 0;
    if (a >= n) return 0;
    uint64_t end = a + SMH_SEG;
    if (end > n) end = n;
    if constexpr (KIND == 5) {
        /* from the definition: END column e is a candidate when the 8-gram that ends at every ODD column c in (e - J, e]
         * is in plane e - c (a lookup without eight symbols in front of it cannot rule anything out) */
        const int J = P.gram_planes;
        uint32_t cnt5 = 0;
        for (uint64_t e = a; e < end; ++e) {
            if (e + 1 < (uint64_t)P.m) continue;
            bool cand = true;
            for (int64_t c = (int64_t)(e | 1u) - ((e & 1u) ? 0 : 2); cand && c >= 7 && (int64_t)e - c < (int64_t)J; c -= 2) {
                uint32_t code = 0;
                for (int i = 7; i >= 0; --i) code = (code << 2) | (text[(uint64_t)c - (uint64_t)i] & 3u);
                if ((smh_lds_u16(tab, 2u * code) >> (J - 1 - (int)((int64_t)e - c))) & 1u) cand = false;
            }
            if (cand) {
                const uint32_t hit = smh_wm_verify(text, e, P);
                cnt5 += hit;
                if (match_mask && hit) *match_mask |= 1ull << (e - a);
            }
        }
        return cnt5;
    }
    const uint64_t q = KIND == 1 ? 7u : (KIND == 3 ? 8u : (smh_kind_wide(KIND) ? 4u : 3u));
    const uint32_t cand_bit = KIND == 1 ? (uint32_t)P.gram_planes - 1u : 7u;
    uint32_t T = KIND == 1 ? smh_gram1_state_before(text, a, P.gram_g7, P.gram_planes)
                           : smh_gram_state_before<KIND>(text, a, tab, P.gram_g7, 0xFFu & ~((1u << (8 - (KIND == 1 ? 8 : P.gram_planes))) - 1u), smh_kind_flat_k2(KIND)), cnt = 0;
    for (uint64_t e = a; e < end; ++e) {
        uint32_t G = 0u; /* a column without a whole gram in front of it cannot be ruled out */
        if (e + 1 >= q) {
            if (KIND == 3) {
                uint32_t code = 0;
                for (int i = 7; i >= 0; --i) code = (code << 2) | (text[e - (uint64_t)i] & 3u);
                G = smh_lds_u8(tab, code);
            } else if (KIND == 1) {
                uint32_t code = 0;
                for (int i = 6; i >= 0; --i) code = (code << 2) | (text[e - (uint64_t)i] & 3u);
                uint16_t g;
                memcpy(&g, P.gram_g7 + 2u * code, 2);
                G = g;
            } else {
                const uint32_t key = smh_kind_wide(KIND) ? (uint32_t)text[e - 3] | ((uint32_t)text[e - 2] << 8) | ((uint32_t)text[e - 1] << 16) | ((uint32_t)text[e] << 24)
                                                         : (uint32_t)text[e - 2] | ((uint32_t)text[e - 1] << 8) | ((uint32_t)text[e] << 16);
                G = smh_gram_byte_G<smh_kind_lookup(KIND)>(key, tab, 0xFFu & ~((1u << (8 - P.gram_planes)) - 1u), smh_kind_flat_k2(KIND));
            }
        }
        T = smh_gram_step(T, G);
        if (!((T >> cand_bit) & 1u) && e + 1 >= (uint64_t)P.m) {
            const uint32_t hit = smh_wm_verify(text, e, P);
            cnt += hit;
            if (match_mask && hit) *match_mask |= 1ull << (e - a);
        }
    }
    return cnt;
}

template <int KIND, bool POS = false, int STG = 0, bool QD = true>
SMH_LANE uint32_t smh_wm_gram_thread(uint64_t gthread, const smh_chunk_sched &S, const uint8_t *text, uint64_t n,
                                     const void *tab, const smh_wm_params &P, uint64_t *queue_base, const smh_pos_out *po = nullptr,
                                     uint32_t smh_gram_drain_at = 64u, const smh_wm_queue *stage = nullptr,
                                     uint32_t *events_out = nullptr)
{
    if (n < (uint64_t)P.m) return 0;
    const uint64_t chunk_bytes = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (n + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    smh_wm_queue Q = {};
    if (stage) Q = *stage; /* staged verify (STG > 0): where the locks, buffers and this wave's list live */
    Q.slots = queue_base;
    Q.count = 0;
    Q.matches = 0;
    Q.events = 0;
    Q.po = POS ? po : nullptr;
    Q.pa_limit = n;
    uint32_t cnt = 0;
    constexpr int HP = smh_stg_hp(STG), HD = 4 * HP; /* 16-byte pieces / dwords of text kept from in front of the chunk */
    [[maybe_unused]] constexpr bool RV = smh_stg_regv(STG);
    uint32_t cur[16], nxt[16], cur_halo[HD], nxt_halo[HD];
    uint64_t k = S.take(n_chunks);
    /* the fast path needs eight bytes in front of the chunk and, for the columns to have full windows, m - 1 of
     * them: chunk 0 is excluded (m - 1 <= 4095 is checked by the launcher); the last chunk only when partial */
    auto is_fast = [&](uint64_t kk) { return kk >= 1 && kk < n_chunks && (kk + 1) * chunk_bytes <= n; };
    auto load = [&](uint64_t kk, uint32_t (&w)[16], uint32_t (&halo)[HD]) {
        const uint64_t base = smh_uniform64(kk * chunk_bytes);
        const uint8_t *p = text + base + (uint64_t)lane * SMH_SEG;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const smh_u32x4 t = smh_load16(p + 16u * q);
            w[4 * q + 0] = t.v[0];
            w[4 * q + 1] = t.v[1];
            w[4 * q + 2] = t.v[2];
            w[4 * q + 3] = t.v[3];
        }
        /* the bytes in front of the wave-chunk, same address in every lane */
#pragma unroll
        for (int q = 0; q < HP; ++q) {
            const smh_u32x4 t = smh_load16(text + base - 16u * (uint32_t)(HP - q));
            halo[4 * q + 0] = t.v[0];
            halo[4 * q + 1] = t.v[1];
            halo[4 * q + 2] = t.v[2];
            halo[4 * q + 3] = t.v[3];
        }
    };
    bool cur_fast = is_fast(k);
    if (cur_fast) load(k, cur, cur_halo);
    while (k < n_chunks) {
        const uint64_t kn = S.take(n_chunks);
        const bool nxt_fast = is_fast(kn);
        /* the verify stage runs HERE, between chunks and before the next chunk's text is requested: its loads
         * return in order behind everything the wave has in flight, so a drain entered while a prefetch is
         * outstanding also waits for that prefetch (measured: 2-3 x the cost per surviving column) */
        if (QD && Q.count >= smh_gram_drain_at) smh_wm_drain(Q, text, P);
        /* staged / in-register verify, pipelined: the buckets of the columns hashed at the end of the last chunk are requested
         * now and looked at after this chunk's scan (by the next flush, or below) */
        if constexpr (RV) smh_wm_pend_issue_rv(Q, P);
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
        else if (STG > 0 && !smh_stg_l2(STG)) smh_wm_pend_issue<false>(Q, P);
#endif
        if (SMH_PREFETCH && nxt_fast) load(kn, nxt, nxt_halo);
        const uint64_t a = smh_uniform64(k * chunk_bytes) + (uint64_t)lane * SMH_SEG;
        if (cur_fast) {
            smh_wm_gram_lane_fast<KIND, POS, STG, QD>(text, a, cur, cur_halo, tab, P, Q);
        } else if constexpr (KIND == 4) {
            cnt += smh_wm_gram2_lane_slow(text, n, a, P, POS ? po : nullptr); /* positions: a column is appended once per pattern that ends there */
        } else if (POS) {
            uint64_t mm;
            smh_wm_gram_lane_slow<KIND>(text, n, a, tab, P, &mm);
            cnt += smh_append_bits(mm, a, *po);
        } else {
            cnt += smh_wm_gram_lane_slow<KIND>(text, n, a, tab, P);
        }
        if constexpr (RV) { if (Q.pend_loaded) smh_wm_pend_finish_rv(Q, text, P); } /* a chunk without a flush of its own */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
        else if constexpr (smh_stg_l2(STG)) { /* a chunk without survivors of its own still moves the pipeline on */
            if (!Q.pend_mine && (Q.pend_n || Q.pa_n)) smh_wm_l2_step<smh_stg_l2_maxd(STG)>(Q, text, 0, P, false);
            Q.pend_mine = 0u;
        } else if (STG > 0 && Q.pend_loaded) smh_wm_pend_finish<false>(Q, text, P);
#endif
        if (nxt_fast) {
            if (SMH_PREFETCH) {
#pragma unroll
                for (int q = 0; q < 16; ++q) cur[q] = nxt[q];
#pragma unroll
                for (int q = 0; q < HD; ++q) cur_halo[q] = nxt_halo[q];
            } else {
                load(kn, cur, cur_halo);
            }
        }
        cur_fast = nxt_fast;
        k = kn;
    }
    if constexpr (RV) smh_wm_pend_finish_rv(Q, text, P);
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    else if constexpr (smh_stg_l2(STG)) { /* empty the queue, then run the pipeline dry */
        while (Q.count) smh_wm_l2_step<smh_stg_l2_maxd(STG)>(Q, text, 0, P, true);
        smh_wm_l2_step<smh_stg_l2_maxd(STG)>(Q, text, 0, P, false);
        smh_wm_l2_step<smh_stg_l2_maxd(STG)>(Q, text, 0, P, false);
    } else if (STG > 0) smh_wm_pend_finish<false>(Q, text, P);
#endif
    if (QD) smh_wm_drain(Q, text, P);
    if (events_out) *events_out = Q.events;
    return cnt + Q.matches;
}

/* ------------------------------------------------------------------ match positions
 * END columns of all matches (wu/wu.c:93 printed "Match of pattern index %i at %i" for the end
 * column, in commented-out code).  One lane tests the 64 end columns of a segment with the
 * reference-layout tables as given (SHIFT skip loop, bucket scan, byte compare) and the wave
 * compacts the hits into the output buffer (smh_append_positions semantics, see ac_lane.h).
 */
template <typename SHIFT_T>
SMH_LANE uint64_t smh_wm_segment_match_mask(const uint8_t *text, uint64_t n, uint64_t a, const SHIFT_T *shift,
                                            uint32_t shiftsize, const uint32_t *bucket_off, const int32_t *bucket,
                                            const uint8_t *pat_orig, int m, int nbits)
{
    uint64_t end = a + SMH_SEG;
    if (end > n) end = n;
    uint64_t column = a;
    if (column < (uint64_t)(m - 1)) column = (uint64_t)(m - 1);
    uint64_t mask = 0;
    while (column < end) {
        uint32_t hash1 = text[column - 2];
        hash1 <<= nbits;
        hash1 += text[column - 1];
        hash1 <<= nbits;
        hash1 += text[column];
        const uint32_t sh = hash1 < shiftsize ? shift[hash1] : 1u;
        if (sh == 0) {
            uint32_t hash2 = text[column - (uint64_t)m + 1];
            hash2 <<= nbits;
            hash2 += text[column - (uint64_t)m + 2];
            const uint32_t b0 = bucket_off[hash1], b1 = bucket_off[hash1 + 1];
            for (uint32_t k = b0; k < b1; ++k) {
                if ((uint32_t)bucket[2 * k] != hash2) continue;
                const uint8_t *q = pat_orig + (uint64_t)(uint32_t)bucket[2 * k + 1] * (uint32_t)m;
                const uint8_t *w = text + (column + 1 - (uint64_t)m);
                int i = 0;
                while (i < m && q[i] == w[i]) ++i;
                if (i == m) {
                    mask |= 1ull << (column - a);
                    break;
                }
            }
            ++column;
        } else {
            column += sh;
        }
    }
    return mask;
}

template <typename SHIFT_T>
SMH_LANE void smh_wm_positions_thread(uint64_t gthread, uint64_t nthreads, const uint8_t *text, uint64_t n,
                                      const SHIFT_T *shift, uint32_t shiftsize, const uint32_t *bucket_off,
                                      const int32_t *bucket, const uint8_t *pat_orig, int m, int nbits,
                                      uint64_t *positions, uint64_t capacity, uint64_t *cursor)
{
    if (n < (uint64_t)m) return;
    const uint64_t n_segs = (n + SMH_SEG - 1) / SMH_SEG;
    const uint64_t n_rounds = (n_segs + nthreads - 1) / nthreads;
    for (uint64_t r = 0; r < n_rounds; ++r) {
        const uint64_t a = (r * nthreads + gthread) * SMH_SEG;
        uint64_t mask = a < n ? smh_wm_segment_match_mask<SHIFT_T>(text, n, a, shift, shiftsize, bucket_off, bucket,
                                                                  pat_orig, m, nbits)
                              : 0ull;
        /* bit b = END column a + b */
        const uint32_t mine = (uint32_t)__builtin_popcountll(mask);
        uint64_t slot = smh_wave_reserve(cursor, mine);
        while (mask) {
            const int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            if (slot < capacity) positions[slot] = a + (uint64_t)b;
            ++slot;
        }
    }
}

#endif
