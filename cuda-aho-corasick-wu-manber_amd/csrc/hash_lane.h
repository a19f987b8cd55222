/*
 * csrc/hash_lane.h -- lane code of the window-hash engine (hash_engine.h): what one lane does with its 64-byte text segment.
 *
 * Stage 1.  A lane owns the 64 END columns of its segment.  Besides the segment it loads the 32 bytes in front of it (priming:
 * the hash of the window that ends just before the segment) and the 68 bytes that start ceil4(m) bytes in front of it -- the
 * OUT stream: the byte that leaves the window at column c is text[c - m], m a run-time value, so the stream is brought to a
 * static position with one v_alignbyte per dword (shift (4 - m) & 3, wave-uniform) instead of indexing registers dynamically.  No
 * cross-lane traffic; the extra loads hit the lines the neighbouring lanes stream anyway.  Per column: v_bfe + v_mad_u32_u24 for
 * the byte that enters, the same for the byte that leaves, two ops for the filter word's address, ds_read_b32, four ops for the
 * two bit tests, one v_alignbit into the lane's candidate mask.
 * Stage 2.  The wave compacts its candidate columns (ballot / mbcnt, a queue of 32-bit chunk offsets in LDS), every lane takes
 * one, requests the window's aligned dwords from global memory (wm_lane.h smh_wm_l2_request), hashes them (smh_wm_tag_dwords),
 * requests BOTH cuckoo slots of the pattern table and compares them with the window in registers.
 *
 * Compiled for the GPU (hash_kernels.hip) and, with SMH_HOST_EMU, for the CPU lane emulator (tests/emu).
 */
#ifndef SMH_HASH_LANE_H
#define SMH_HASH_LANE_H

#include "lane_common.h"
#include "wm_lane.h" /* smh_wm_l2_request, smh_wm_tag_dwords, smh_alignbyte */
#include "hash_engine.h"

#define SMH_HASH_QCAP 192u /* queued candidate columns per wave (32-bit chunk offsets: 768 bytes of LDS) */

struct smh_hash_ctx {
    const uint8_t *text;
    uint64_t n;
    smh_hash_params P;
    const uint8_t *table; /* device memory: 2 * P.slots slots */
};

/* is the m-byte window that ends at column e a stored pattern?  Two dependent round trips: the window, its two slots. */
SMH_LANE uint32_t smh_hash_verify(const smh_hash_ctx &C, uint64_t e, bool wide)
{
    uint32_t d[10];
    const uint32_t sh = smh_wm_l2_request<9>(C.text, e, C.P.m, d, wide);
    const uint32_t tag = smh_wm_tag_dwords<9>(d, sh, C.P.m);
    uint32_t s1, s2;
    smh_hash_slots(tag, C.P.seed, C.P.slots, &s1, &s2);
    const int nd = (C.P.m + 3) >> 2;
    const uint32_t last_mask = (C.P.m & 3) ? (1u << (8 * (C.P.m & 3))) - 1u : 0xFFFFFFFFu;
    uint32_t diff1 = 0, diff2 = 0;
    if (C.P.slot_dwords == 4u) {
        const smh_u32x4 a = smh_load16(C.table + 16u * (uint64_t)s1), b = smh_load16(C.table + 16u * (uint64_t)s2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nd) {
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
                uint32_t v = __builtin_amdgcn_alignbit(d[j + 1], d[j], sh);
#else
                uint32_t v = (uint32_t)((((uint64_t)d[j + 1] << 32) | d[j]) >> sh);
#endif
                if (j == nd - 1) v &= last_mask;
                diff1 |= v ^ a.v[j];
                diff2 |= v ^ b.v[j];
            }
        }
    } else {
        const smh_u32x4 a0 = smh_load16(C.table + 32u * (uint64_t)s1), a1 = smh_load16(C.table + 32u * (uint64_t)s1 + 16u);
        const smh_u32x4 b0 = smh_load16(C.table + 32u * (uint64_t)s2), b1 = smh_load16(C.table + 32u * (uint64_t)s2 + 16u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < nd) {
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
                uint32_t v = __builtin_amdgcn_alignbit(d[j + 1], d[j], sh);
#else
                uint32_t v = (uint32_t)((((uint64_t)d[j + 1] << 32) | d[j]) >> sh);
#endif
                if (j == nd - 1) v &= last_mask;
                diff1 |= v ^ (j < 4 ? a0.v[j] : a1.v[j - 4]);
                diff2 |= v ^ (j < 4 ? b0.v[j] : b1.v[j - 4]);
            }
        }
    }
    return ((diff1 == 0u) | (diff2 == 0u)) ? 1u : 0u;
}

/* per-wave state of stage 2 */
struct smh_hash_queue {
    uint32_t *slots;   /* SMH_HASH_QCAP chunk offsets, private to this wave (LDS on the GPU) */
    uint32_t count;    /* wave-uniform */
    uint32_t matches;  /* per lane */
    uint32_t events;   /* per lane: columns this lane sent to stage 2 (smh_stats.h) */
    const smh_pos_out *po;
};

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
/* verify queue entries [0, Q.count) of the wave-chunk at chunk_base, 64 at a time; all 64 lanes call it */
SMH_LANE void smh_hash_drain(smh_hash_queue &Q, const smh_hash_ctx &C, uint64_t chunk_base)
{
    const uint32_t lane = threadIdx.x & 63u;
    /* 40 bytes from a window's first aligned dword on lie inside the text for every window of this chunk (smh_wm_l2_request "wide") */
    const bool wide = chunk_base + 4096u + 40u <= C.n;
    for (uint32_t base = 0; base < Q.count; base += 64u) {
        const bool mine = base + lane < Q.count;
        const uint64_t e = chunk_base + Q.slots[mine ? base + lane : 0u]; /* a lane without an entry re-checks entry 0 and drops the answer */
        const uint32_t r = smh_hash_verify(C, e, wide);
        Q.matches += mine ? r : 0u;
        if (Q.po) smh_append_bits(mine ? r : 0u, e, *Q.po);
    }
    Q.count = 0u;
}
/* the candidate columns `msk` (bit b = column a + b) of a lane's segment: queued, drained whenever 64 slots might not be free */
SMH_LANE void smh_hash_columns(smh_hash_queue &Q, const smh_hash_ctx &C, uint64_t chunk_base, uint64_t a, uint64_t msk)
{
    Q.events += (uint32_t)__builtin_popcountll(msk);
    while (SMH_WAVE_ANY(msk != 0)) {
        if (Q.count + 64u > SMH_HASH_QCAP) smh_hash_drain(Q, C, chunk_base);
        const bool have = msk != 0;
        const uint32_t b = have ? (uint32_t)__builtin_ctzll(msk) : 0u;
        const uint64_t mask = __ballot(have);
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (have) Q.slots[Q.count + before] = (uint32_t)(a - chunk_base) + b;
        Q.count += (uint32_t)__popcll(mask);
        msk &= msk - 1u;
    }
    smh_hash_drain(Q, C, chunk_base);
}
#else
/* CPU emulation (one lane at a time): the same verify per candidate column */
SMH_LANE void smh_hash_columns(smh_hash_queue &Q, const smh_hash_ctx &C, uint64_t, uint64_t a, uint64_t msk)
{
    Q.events += (uint32_t)__builtin_popcountll(msk);
    while (msk) {
        const uint64_t e = a + (uint64_t)__builtin_ctzll(msk);
        const uint32_t hit = smh_hash_verify(C, e, false);
        Q.matches += hit;
        if (hit && Q.po) smh_append_bits(1u, e, *Q.po);
        msk &= msk - 1u;
    }
}
#endif

/* the filter's answer for the window whose rolling hash is h */
SMH_LANE uint32_t smh_hash_test(uint32_t h, const void *bloom, const smh_hash_params &P)
{
    const uint32_t word = smh_lds_u32(bloom, smh_hash_word_addr(h, P.bloom_shift, P.bloom_mask));
    return smh_bit_at(word, h) & smh_bit_at(word, h >> 5);
}

/* stage 1, fast path: the candidate mask of the 64 END columns of the segment at a (a >= 64, a + 64 <= n).
 * w = the segment, halo = the 32 bytes in front of it, o = the 17 aligned dwords from a - ceil4(m) on. */
SMH_LANE uint64_t smh_hash_lane_fast(const uint32_t (&w)[16], const uint32_t (&halo)[8], const uint32_t (&o)[17], const void *bloom,
                                     const smh_hash_params &P)
{
    const uint32_t m = (uint32_t)P.m;
    /* priming: Horner over the halo's last m bytes -- the bytes in front of them are cleared first (wave-uniform selects) */
    uint32_t h = 0;
    const uint32_t first = 32u - m; /* offset of the window's first byte in the halo (m <= 32) */
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint32_t v = halo[q];
        const uint32_t lo = 4u * (uint32_t)q;
        if (lo + 4u <= first) v = 0u;                                   /* wholly in front of the window */
        else if (lo < first) v &= 0xFFFFFFFFu << (8u * (first - lo));   /* the window starts inside this dword */
#pragma unroll
        for (int k = 0; k < 4; ++k) h = smh_hash_in(h, smh_bfe(v, 8u * k, 8u));
    }
    /* h = hash of text[a - m, a): the window that ends at column a - 1 */
    const uint32_t osh = (4u - (m & 3u)) & 3u; /* byte offset of text[a - m] in o[0] */
    uint32_t mlo = 0, mhi = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t out = smh_alignbyte(o[q + 1], o[q], osh); /* text[a - m + 4q .. +3] */
        uint32_t hh[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h = smh_hash_in(h, smh_bfe(w[q], 8u * k, 8u));
            h = smh_hash_out(h, smh_bfe(out, 8u * k, 8u), P.neg_bm);
            hh[k] = h;
        }
        uint32_t bits = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) bits |= smh_hash_test(hh[k], bloom, P) << k;
        if (q < 8) mlo |= bits << (4 * q);
        else mhi |= bits << (4 * (q - 8));
    }
    return ((uint64_t)mhi << 32) | mlo;
}

/* bounds-checked path: the END columns [max(a, m - 1), min(a + 64, n)) byte by byte from memory, filter and verify */
SMH_LANE uint64_t smh_hash_lane_slow(const smh_hash_ctx &C, uint64_t a, const void *bloom)
{
    if (a >= C.n) return 0;
    uint64_t end = a + SMH_SEG;
    if (end > C.n) end = C.n;
    uint64_t e0 = a;
    const uint64_t m = (uint64_t)C.P.m;
    if (e0 < m - 1) e0 = m - 1;
    if (e0 >= end) return 0;
    uint32_t h = 0;
    for (uint64_t i = e0 + 1 - m; i < e0; ++i) h = smh_hash_in(h, C.text[i]);
    uint64_t msk = 0;
    for (uint64_t e = e0; e < end; ++e) {
        h = smh_hash_in(h, C.text[e]);
        if (e > e0) h = smh_hash_out(h, C.text[e - m], C.P.neg_bm); /* (the first window was primed with exactly its own bytes) */
        if (smh_hash_test(h, bloom, C.P)) msk |= 1ull << (e - a);
    }
    return msk;
}

/* a candidate column of the bounds-checked path: its window may touch the text's first / last bytes, so it is compared byte by byte */
SMH_LANE uint32_t smh_hash_verify_bytes(const smh_hash_ctx &C, uint64_t e)
{
    const int m = C.P.m;
    uint32_t d[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < m; ++i) d[i >> 2] |= (uint32_t)C.text[e + 1 - (uint64_t)m + (uint64_t)i] << (8 * (i & 3));
    uint32_t tag = 0x811C9DC5u;
    for (int j = 0; j < (m + 3) >> 2; ++j) tag = smh_wm_mix(tag, d[j]);
    uint32_t s1, s2;
    smh_hash_slots(tag, C.P.seed, C.P.slots, &s1, &s2);
    const uint32_t sb = 4u * C.P.slot_dwords;
    uint32_t diff1 = 0, diff2 = 0;
    for (uint32_t j = 0; j < C.P.slot_dwords; ++j) {
        uint32_t a, b;
        memcpy(&a, C.table + (uint64_t)s1 * sb + 4u * j, 4);
        memcpy(&b, C.table + (uint64_t)s2 * sb + 4u * j, 4);
        diff1 |= a ^ d[j];
        diff2 |= b ^ d[j];
    }
    return ((diff1 == 0u) | (diff2 == 0u)) ? 1u : 0u;
}

template <bool POS>
SMH_LANE uint32_t smh_hash_thread(uint64_t gthread, const smh_chunk_sched &S, const smh_hash_ctx &C, const void *bloom, uint32_t *queue,
                                  const smh_pos_out *po, uint32_t *events_out)
{
    if (C.n < (uint64_t)C.P.m) return 0;
    const uint64_t chunk_bytes = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (C.n + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    smh_hash_queue Q = {};
    Q.slots = queue;
    Q.po = POS ? po : nullptr;
    const uint32_t D = ((uint32_t)C.P.m + 3u) & ~3u; /* the out stream starts D bytes in front of the segment */
    /* fast chunks: text in front (chunk >= 1) and 16 bytes behind the chunk's last window dword inside the text */
    auto is_fast = [&](uint64_t kk) { return kk >= 1 && kk < n_chunks && (kk + 1) * chunk_bytes + 64u <= C.n; };
    uint64_t k = S.take(n_chunks);
    while (k < n_chunks) {
        const uint64_t chunk_base = smh_uniform64(k * chunk_bytes);
        const uint64_t a = chunk_base + (uint64_t)lane * SMH_SEG;
        if (is_fast(k)) {
            uint32_t w[16], halo[8], o[17];
            const uint8_t *p = C.text + a;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const smh_u32x4 t = smh_load16(p + 16u * q);
                w[4 * q + 0] = t.v[0]; w[4 * q + 1] = t.v[1]; w[4 * q + 2] = t.v[2]; w[4 * q + 3] = t.v[3];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const smh_u32x4 t = smh_load16(p - 32u + 16u * q);
                halo[4 * q + 0] = t.v[0]; halo[4 * q + 1] = t.v[1]; halo[4 * q + 2] = t.v[2]; halo[4 * q + 3] = t.v[3];
            }
            const uint32_t *op = reinterpret_cast<const uint32_t *>(p - D);
#pragma unroll
            for (int q = 0; q < 17; ++q) o[q] = op[q];
            const uint64_t msk = smh_hash_lane_fast(w, halo, o, bloom, C.P);
            smh_hash_columns(Q, C, chunk_base, a, msk);
        } else {
            uint64_t msk = smh_hash_lane_slow(C, a, bloom);
            Q.events += (uint32_t)__builtin_popcountll(msk);
            uint64_t hits = 0;
            while (msk) {
                const int b = __builtin_ctzll(msk);
                msk &= msk - 1u;
                if (smh_hash_verify_bytes(C, a + (uint64_t)b)) hits |= 1ull << b;
            }
            Q.matches += (uint32_t)__builtin_popcountll(hits);
            if (POS) smh_append_bits(hits, a, *po);
        }
        k = S.take(n_chunks);
    }
    if (events_out) *events_out = Q.events;
    return Q.matches;
}

#endif
