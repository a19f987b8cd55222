/*
 * csrc/hash_lane.h -- lane code of the window-hash engine (hash_engine.h): what one lane does with its 64-byte text segment.
 *
 * Stage 1.  A lane owns the 64 END columns of its segment.  The 32 bytes in front of it (priming: the hash of the window that
 * ends just before the segment) are the previous lane's last eight registers (DPP wave_shr:1; lane 0: the wave-uniform bytes in
 * front of the wave-chunk).  The byte that leaves the window at column c is text[c - m]: the OUT stream, the 68 bytes from
 * ceil4(m) = 4 ND bytes in front of the segment on -- the halo's last ND dwords and the segment's first 17 - ND (ND a template
 * value: static register positions), brought to byte alignment with one v_alignbyte per dword (shift (4 - m) & 3).  Per column: v_bfe + v_mad_u32_u24 for
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

#define SMH_HASH_QCAP 320u /* queued candidate columns per wave (32-bit chunk offsets: 1280 bytes of LDS): a drain takes up to 256 as 128 + 128 */

struct smh_hash_ctx {
    const uint8_t *text;
    uint64_t n;
    smh_hash_params P;
    const uint8_t *table; /* device memory: 2 * P.slots slots */
    uint32_t drop;        /* development knob SMH_HASH_TUNE="drop=1": candidate columns are not verified (what stage 1 alone costs; counts are wrong) */
};

/* the window's dwords, END-aligned to a dword boundary or not: dword j of the m-byte window that starts `sh` bits into d[0] */
SMH_LANE uint32_t smh_hash_window_dword(const uint32_t (&d)[10], uint32_t sh, int j)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    return __builtin_amdgcn_alignbit(d[j + 1], d[j], sh);
#else
    return (uint32_t)((((uint64_t)d[j + 1] << 32) | d[j]) >> sh);
#endif
}

/* a bucket's N dwords (two slots interleaved, 4-byte aligned, N = 2 * ND even) in as few requests as their number allows */
template <int N>
SMH_LANE void smh_hash_load_bucket(const uint8_t *p, uint32_t (&v)[N])
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    typedef uint32_t v4a __attribute__((ext_vector_type(4), aligned(4)));
    typedef uint32_t v2a __attribute__((ext_vector_type(2), aligned(4)));
    const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int g = 0; 4 * g < N; ++g) {
        if constexpr (true) {
            if (4 * g + 4 <= N) {
                const v4a t = *reinterpret_cast<const v4a *>(q + 4 * g);
                v[4 * g] = t.x; v[4 * g + 1] = t.y; v[4 * g + 2] = t.z; v[4 * g + 3] = t.w;
            } else {
                const v2a t = *reinterpret_cast<const v2a *>(q + 4 * g);
                v[4 * g] = t.x; v[4 * g + 1] = t.y;
            }
        }
    }
#else
    for (int j = 0; j < N; ++j) memcpy(&v[j], p + 4 * j, 4);
#endif
}

/* the aligned dwords of the m-byte window that ends at column e, as 16-byte requests: ONE covers a window of up to 13 bytes at any
 * alignment, two up to 29, three the rest -- every request of a wave touches 64 different cache lines, and the stage is bound by
 * the rate at which the L1 takes them (3 + 2 requests per window and slot pair measured 1.8 ms/GiB at 415 windows per 4 KiB).
 * The caller guarantees 48 bytes from the window's first aligned dword on inside the text (`wide`); else the narrow form. */
SMH_LANE uint32_t smh_hash_request(const uint8_t *text, uint64_t e, int m, uint32_t (&d)[10], bool wide)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    if (wide) {
        const uint64_t s0 = e + 1 - (uint64_t)m;
        const uint8_t *al = text + (s0 & ~(uint64_t)3);
        typedef uint32_t v4a __attribute__((ext_vector_type(4), aligned(4)));
        const v4a q0 = *reinterpret_cast<const v4a *>(al);
        d[0] = q0.x; d[1] = q0.y; d[2] = q0.z; d[3] = q0.w;
        if (m > 13) { /* wave-uniform */
            const v4a q1 = *reinterpret_cast<const v4a *>(al + 16);
            d[4] = q1.x; d[5] = q1.y; d[6] = q1.z; d[7] = q1.w;
        } else {
            d[4] = d[5] = d[6] = d[7] = 0u;
        }
        if (m > 29) {
            typedef uint32_t v2a __attribute__((ext_vector_type(2), aligned(4)));
            const v2a q2 = *reinterpret_cast<const v2a *>(al + 32);
            d[8] = q2.x; d[9] = q2.y;
        } else {
            d[8] = d[9] = 0u;
        }
        return (uint32_t)(s0 & 3u) * 8u;
    }
#endif
    return smh_wm_l2_request<9>(text, e, m, d, false);
}

/* ND = (m + 3) / 4, the window's dwords: a template value so that every register array below has a static shape (indexed by
 * a run-time m the compiler keeps them in scratch memory) */
template <int NCH, int ND>
SMH_LANE void smh_hash_verify(const smh_hash_ctx &C, const uint64_t (&e)[NCH], bool wide, uint32_t (&hit)[NCH])
{
    uint32_t d[NCH][10], sh[NCH], s1[NCH], s2[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) sh[c] = smh_hash_request(C.text, e[c], C.P.m, d[c], wide);
#pragma unroll
    for (int c = 0; c < NCH; ++c) smh_hash_slots(smh_wm_tag_dwords<9>(d[c], sh[c], C.P.m), C.P.seed, C.P.slots, &s1[c], &s2[c]);
    const uint32_t last_mask = (C.P.m & 3) ? (1u << (8 * (C.P.m & 3))) - 1u : 0xFFFFFFFFu;
    uint32_t a[NCH][2 * ND], b[NCH][2 * ND]; /* the two buckets, each two slots interleaved dword by dword */
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        smh_hash_load_bucket<2 * ND>(C.table + 8u * ND * (uint64_t)s1[c], a[c]);
        smh_hash_load_bucket<2 * ND>(C.table + 8u * ND * (uint64_t)s2[c], b[c]);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        uint32_t d0 = 0, d1 = 0, d2 = 0, d3 = 0;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            uint32_t v = smh_hash_window_dword(d[c], sh[c], j);
            if (j == ND - 1) v &= last_mask;
            /* a bucket's two slots are interleaved dword by dword: slot k's dword j is register 2 j + k, a static position */
            d0 |= v ^ a[c][2 * j];
            d1 |= v ^ a[c][2 * j + 1];
            d2 |= v ^ b[c][2 * j];
            d3 |= v ^ b[c][2 * j + 1];
        }
        hit[c] = ((d0 == 0u) | (d1 == 0u) | (d2 == 0u) | (d3 == 0u)) ? 1u : 0u;
    }
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
template <int ND>
SMH_LANE void smh_hash_drain(smh_hash_queue &Q, const smh_hash_ctx &C, uint64_t chunk_base)
{
    const uint32_t lane = threadIdx.x & 63u;
    /* 48 bytes from a window's first aligned dword on lie inside the text for every window of this chunk (smh_hash_request "wide") */
    const bool wide = chunk_base + 4096u + 48u <= C.n;
    uint32_t base = 0;
    for (; base + 64u < Q.count; base += 128u) { /* two windows per lane: entries base + lane and base + 64 + lane */
        const bool mine1 = base + 64u + lane < Q.count;
        const uint64_t e[2] = {chunk_base + Q.slots[base + lane], chunk_base + Q.slots[mine1 ? base + 64u + lane : 0u]};
        uint32_t r[2];
        smh_hash_verify<2, ND>(C, e, wide, r);
        Q.matches += r[0] + (mine1 ? r[1] : 0u);
        if (Q.po) {
            smh_append_bits(r[0], e[0], *Q.po);
            smh_append_bits(mine1 ? r[1] : 0u, e[1], *Q.po);
        }
    }
    if (base < Q.count) { /* at most 64 left */
        const bool mine = base + lane < Q.count;
        const uint64_t e[1] = {chunk_base + Q.slots[mine ? base + lane : 0u]}; /* a lane without an entry re-checks entry 0 and drops the answer */
        uint32_t r[1];
        smh_hash_verify<1, ND>(C, e, wide, r);
        Q.matches += mine ? r[0] : 0u;
        if (Q.po) smh_append_bits(mine ? r[0] : 0u, e[0], *Q.po);
    }
    Q.count = 0u;
}
/* the candidate columns `msk` (bit b = column a + b) of a lane's segment: queued, drained whenever 64 slots might not be free */
template <int ND>
SMH_LANE void smh_hash_columns(smh_hash_queue &Q, const smh_hash_ctx &C, uint64_t chunk_base, uint64_t a, uint64_t msk)
{
    Q.events += (uint32_t)__builtin_popcountll(msk);
    if (C.drop) return;
    while (SMH_WAVE_ANY(msk != 0)) {
        if (Q.count + 64u > SMH_HASH_QCAP) smh_hash_drain<ND>(Q, C, chunk_base);
        const bool have = msk != 0;
        const uint32_t b = have ? (uint32_t)__builtin_ctzll(msk) : 0u;
        const uint64_t mask = __ballot(have);
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (have) Q.slots[Q.count + before] = (uint32_t)(a - chunk_base) + b;
        Q.count += (uint32_t)__popcll(mask);
        msk &= msk - 1u;
    }
    smh_hash_drain<ND>(Q, C, chunk_base);
}
#else
/* CPU emulation (one lane at a time): the same verify per candidate column */
template <int ND>
SMH_LANE void smh_hash_columns(smh_hash_queue &Q, const smh_hash_ctx &C, uint64_t, uint64_t a, uint64_t msk)
{
    Q.events += (uint32_t)__builtin_popcountll(msk);
    while (msk) {
        const uint64_t e[1] = {a + (uint64_t)__builtin_ctzll(msk)};
        uint32_t hit[1];
        smh_hash_verify<1, ND>(C, e, false, hit);
        Q.matches += hit[0];
        if (hit[0] && Q.po) smh_append_bits(1u, e[0], *Q.po);
        msk &= msk - 1u;
    }
}
#endif

/* the filter's answer for the window whose rolling hash is h */
SMH_LANE uint32_t smh_hash_test(uint32_t h, const void *bloom, const smh_hash_params &P, bool k3)
{
    const uint32_t word = smh_lds_u32(bloom, smh_hash_word_addr(h, P.bloom_shift, P.bloom_mask));
    uint32_t pass = smh_bit_at(word, h) & smh_bit_at(word, h >> P.bit2_shift);
    if (k3) pass &= smh_bit_at(word, smh_hash_bit3_raw(h)); /* round 6: four more vector instructions per column, a third fewer false candidates */
    return pass;
}

/* stage 1, fast path: the candidate mask of the 64 END columns of the segment at a (a >= 64, a + 64 <= n).
 * w = the segment, halo = the 32 bytes in front of it, o = the 17 aligned dwords from a - ceil4(m) on. */
template <bool K3>
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
        for (int k = 0; k < 4; ++k) bits |= smh_hash_test(hh[k], bloom, P, K3) << k;
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
        if (smh_hash_test(h, bloom, C.P, C.P.bloom_k >= 3u)) msk |= 1ull << (e - a);
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
    uint32_t hit = 0;
    for (uint32_t slot = 0; slot < 4u; ++slot) { /* both slots of both buckets */
        const uint8_t *q = C.table + (uint64_t)(slot < 2u ? s1 : s2) * 2u * sb; /* the bucket: slot k's dword j at dword 2 j + k */
        uint32_t diff = 0;
        for (uint32_t j = 0; j < C.P.slot_dwords; ++j) {
            uint32_t a;
            memcpy(&a, q + 4u * (2u * j + (slot & 1u)), 4);
            diff |= a ^ d[j];
        }
        hit |= diff == 0u ? 1u : 0u;
    }
    return hit;
}

/* K3: the fast path tests the filter's third bit (a set built with bloom_k = 3; testing two bits of it is correct as well --
 * stage 2 decides -- and is what the positions kernels do) */
template <bool POS, int ND, bool K3 = false>
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
    /* fast chunks: text in front (chunk >= 1) and 16 bytes behind the chunk's last window dword inside the text */
    auto is_fast = [&](uint64_t kk) { return kk >= 1 && kk < n_chunks && (kk + 1) * chunk_bytes + 64u <= C.n; };
    uint64_t k = S.take(n_chunks);
    while (k < n_chunks) {
        const uint64_t chunk_base = smh_uniform64(k * chunk_bytes);
        const uint64_t a = chunk_base + (uint64_t)lane * SMH_SEG;
        if (is_fast(k)) {
            uint32_t w[16], edge[8], halo[8], o[17];
            const uint8_t *p = C.text + a;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const smh_u32x4 t = smh_load16(p + 16u * q);
                w[4 * q + 0] = t.v[0]; w[4 * q + 1] = t.v[1]; w[4 * q + 2] = t.v[2]; w[4 * q + 3] = t.v[3];
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) { /* the 32 bytes in front of the wave-chunk: the same address in every lane */
                const smh_u32x4 t = smh_load16(C.text + chunk_base - 32u + 16u * q);
                edge[4 * q + 0] = t.v[0]; edge[4 * q + 1] = t.v[1]; edge[4 * q + 2] = t.v[2]; edge[4 * q + 3] = t.v[3];
            }
            /* the 32 bytes in front of the segment are the previous lane's last eight registers (DPP wave_shr:1; lane 0: the edge) ... */
#pragma unroll
            for (int q = 0; q < 8; ++q) halo[q] = smh_prev_lane_word(w[8 + q], edge[q], C.text, a - 32u + 4u * q);
            /* ... and the OUT stream -- the 17 aligned dwords from a - 4 ND on -- is the halo's last ND dwords and the segment's first
             * 17 - ND: static register positions, ND being a template value */
#pragma unroll
            for (int q = 0; q < 17; ++q) o[q] = q < ND ? halo[8 - ND + q] : w[q - ND];
            const uint64_t msk = smh_hash_lane_fast<K3>(w, halo, o, bloom, C.P);
            smh_hash_columns<ND>(Q, C, chunk_base, a, msk);
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
