/*
 * csrc/acm_lane.h -- what one lane of the mixed-length automaton kernel does (acm_host.c describes the automaton).
 *
 * End ownership: the lane counts, for every position of ITS 64-byte segment, the joined output count of the
 * state after that position.  Its state is warmed up over the K-1 bytes in front of the segment (the depth-K
 * automaton's state depends on the last K symbols only), loaded together with the segment.  Candidate bits
 * (the K-symbol prefix of a longer pattern ends here) are recorded one per position and, after the segment,
 * queued per wave (ballot + prefix count) for the walk down the goto trie in HBM.
 */
#ifndef SMH_ACM_LANE_H
#define SMH_ACM_LANE_H

#include "lane_common.h"

#define SMH_ACM_QCAP 256u /* queue entries per wave (HBM workspace, 8 bytes each) */

/* 16-bit entries: candidate << 15 | count << 13 | row.  32-bit entries are laid out for the instruction count of the
 * scan (the kernel was VALU-bound at eight ops per byte): bit 0 = candidate (one v_alignbit shifts it into the
 * lane's candidate mask), bits 3..23 = BYTE OFFSET of the next row in the table (the entry, masked, is the address:
 * v_and_or with the symbol's offset), bits 24..30 = joined output count (added with a byte-select add). */
template <typename E> struct smh_acm_entry;
template <> struct smh_acm_entry<uint16_t> { static constexpr uint32_t ROW = 0x1FFFu, CNT_SHIFT = 13, CNT_BITS = 2, CAND_SHIFT = 15; };
template <> struct smh_acm_entry<uint32_t> { static constexpr uint32_t ROW = 0x00FFFFF8u, CNT_SHIFT = 24, CNT_BITS = 7, CAND_SHIFT = 0; };

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
/* acc + byte 3 of e, one VALU op (SDWA byte select) */
SMH_LANE uint32_t smh_add_byte3(uint32_t acc, uint32_t e)
{
    uint32_t r;
    asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(acc), "v"(e));
    return r;
}
SMH_LANE uint32_t smh_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return __builtin_amdgcn_alignbit(hi, lo, sh); }
#else
SMH_LANE uint32_t smh_add_byte3(uint32_t acc, uint32_t e) { return acc + (e >> 24); }
SMH_LANE uint32_t smh_alignbit(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh); }
#endif

struct smh_acm_ctx { /* wave-uniform */
    const uint8_t *text;
    uint64_t n;
    int K;
    int max_len;
    int sigma;
    const uint32_t *g_goto; /* HBM */
    const uint8_t *g_final; /* HBM */
};

/* a K-symbol prefix of a longer pattern ends at text[q]: follow goto edges from the root over the text that
 * starts K-1 symbols earlier; every pattern end DEEPER than K on that path is one occurrence (the shallower
 * ones were counted by the scan) */
SMH_LANE uint32_t smh_acm_walk(const smh_acm_ctx &C, uint64_t q)
{
    const uint64_t start = q + 1 - (uint64_t)C.K;
    uint32_t node = 0, cnt = 0;
    for (int t = 0; t < C.max_len; ++t) {
        const uint64_t p = start + (uint64_t)t;
        if (p >= C.n) break;
        const uint32_t c = C.text[p];
        if (c >= (uint32_t)C.sigma) break;
        node = C.g_goto[(uint64_t)node * (uint32_t)C.sigma + c];
        if (node == 0) break;
        if (t >= C.K) cnt += C.g_final[node];
    }
    return cnt;
}

struct smh_acm_queue {
    uint64_t *slots; /* SMH_ACM_QCAP entries in HBM, private to this wave */
    uint32_t count;  /* wave-uniform */
    uint32_t matches;
};

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE void smh_acm_drain(smh_acm_queue &Q, const smh_acm_ctx &C)
{
    if (Q.count == 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the entries were written by this wave (ac_lane.h smh_ac_drain) */
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < Q.count; i += 64u) {
        const uint64_t q = __hip_atomic_load(Q.slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        Q.matches += smh_acm_walk(C, q);
    }
    Q.count = 0;
}
SMH_LANE void smh_acm_emit(smh_acm_queue &Q, const smh_acm_ctx &C, bool cond, uint64_t pos)
{
    const uint64_t mask = __ballot(cond);
    if (mask == 0) return;
    const uint32_t np = (uint32_t)__popcll(mask);
    if (Q.count + np > SMH_ACM_QCAP) smh_acm_drain(Q, C);
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (cond) __hip_atomic_store(Q.slots + Q.count + before, pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Q.count += np;
}
#else
SMH_LANE void smh_acm_drain(smh_acm_queue &, const smh_acm_ctx &) {}
SMH_LANE void smh_acm_emit(smh_acm_queue &Q, const smh_acm_ctx &C, bool cond, uint64_t pos)
{
    if (cond) Q.matches += smh_acm_walk(C, pos);
}
#endif

/* one transition: entry after consuming byte k of text dword w (SIGMA == 4: symbols are two bits) */
template <typename E, int SIGMA>
SMH_LANE uint32_t smh_acm_next(uint32_t e, uint32_t w, int k, const void *tab, int sigma_rt)
{
    using X = smh_acm_entry<E>;
    if (sizeof(E) == 4) {
        /* the masked entry is the next row's byte offset; `w` is the text dword << 2 for SIGMA == 4 (smh_acm_prep) */
        if (SIGMA == 4) return smh_lds_u32(tab, (e & X::ROW) | smh_bfe(w, 8 * k, 4));
        uint32_t c = smh_byte_of(w, k);
        if (c >= (uint32_t)sigma_rt) c = 0; /* never an out-of-range index (smatcher_hip.h: text symbols must be < alphabet) */
        return smh_lds_u32(tab, (e & X::ROW) + 4u * c);
    }
    if (SIGMA == 4) {
        const uint32_t c = smh_bfe(w, 8 * k, 2);
        const uint32_t addr = (((e & X::ROW) << 2) | c) * (uint32_t)sizeof(E);
        return smh_lds_u16(tab, addr);
    } else {
        const uint32_t sigma = (uint32_t)sigma_rt;
        uint32_t c = smh_byte_of(w, k);
        if (c >= sigma) c = 0;
        const uint32_t addr = ((e & X::ROW) * sigma + c) * (uint32_t)sizeof(E);
        return smh_lds_u16(tab, addr);
    }
}
template <typename E, int SIGMA> SMH_LANE uint32_t smh_acm_prep(uint32_t w) { return sizeof(E) == 4 && SIGMA == 4 ? w << 2 : w; }

/* fast path: segment at a (a >= 16, a + 64 <= n); w[0..3] = the 16 bytes in front of it, w[4..19] = the segment */
template <typename E, int SIGMA>
SMH_LANE uint32_t smh_acm_lane_fast(uint64_t a, const uint32_t (&w)[20], const void *tab, const smh_acm_ctx &C, smh_acm_queue &Q)
{
    using X = smh_acm_entry<E>;
    uint32_t e = 0, cnt = 0, clo = 0, chi = 0;
    /* warm-up over the K-1 bytes in front of the segment (K - 1 <= 16): no counting, no candidates -- those
     * positions belong to the previous lane */
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t x = smh_acm_prep<E, SIGMA>(w[q]);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * q + k >= 17 - C.K) e = smh_acm_next<E, SIGMA>(e, x, k, tab, C.sigma);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t x = smh_acm_prep<E, SIGMA>(w[4 + q]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * q + k;
            e = smh_acm_next<E, SIGMA>(e, x, k, tab, C.sigma);
            if (sizeof(E) == 4) {
                /* per byte: v_bfe + v_and_or (address), ds_read_b32, one add (count), one v_alignbit (candidate) */
                cnt = smh_add_byte3(cnt, e);
                if (i < 32) clo = smh_alignbit(e, clo, 1u);
                else chi = smh_alignbit(e, chi, 1u);
            } else {
                cnt += smh_bfe(e, X::CNT_SHIFT, X::CNT_BITS);
                if (i < 32) clo |= (e >> X::CAND_SHIFT) << i;
                else chi |= (e >> X::CAND_SHIFT) << (i - 32);
            }
        }
    }
    uint64_t msk = ((uint64_t)chi << 32) | clo;
    while (SMH_WAVE_ANY(msk != 0)) {
        const bool have = msk != 0;
        const uint32_t b = have ? (uint32_t)__builtin_ctzll(msk) : 0u;
        smh_acm_emit(Q, C, have, a + b);
        msk &= msk - 1u;
    }
    return cnt;
}

/* bounds-checked path for the first chunk (no 16 bytes in front) and the last one: byte loads, table in HBM */
template <typename E>
SMH_LANE uint32_t smh_acm_lane_slow(uint64_t a, const void *tab_g, const smh_acm_ctx &C)
{
    using X = smh_acm_entry<E>;
    if (a >= C.n) return 0;
    uint64_t end = a + SMH_SEG;
    if (end > C.n) end = C.n;
    const uint64_t warm = a >= (uint64_t)(C.K - 1) ? a - (uint64_t)(C.K - 1) : 0;
    uint32_t e = 0, cnt = 0;
    for (uint64_t i = warm; i < end; ++i) {
        uint32_t c = C.text[i];
        if (c >= (uint32_t)C.sigma) c = 0;
        if (sizeof(E) == 4) memcpy(&e, (const uint8_t *)tab_g + (e & X::ROW) + 4u * c, 4); /* the masked entry is a byte offset */
        else e = ((const E *)tab_g)[(uint64_t)(e & X::ROW) * (uint32_t)C.sigma + c];
        if (i >= a) {
            cnt += (e >> X::CNT_SHIFT) & ((1u << X::CNT_BITS) - 1u);
            if ((e >> X::CAND_SHIFT) & 1u) cnt += smh_acm_walk(C, i);
        }
    }
    return cnt;
}

template <typename E, int SIGMA>
SMH_LANE uint32_t smh_acm_thread(uint64_t gthread, const smh_chunk_sched &S, const void *tab, const void *tab_g,
                                 const smh_acm_ctx &C, uint64_t *queue_base)
{
    if (C.n == 0) return 0;
    const uint64_t chunk_bytes = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (C.n + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    smh_acm_queue Q;
    Q.slots = queue_base ? queue_base + smh_uniform64(gthread >> 6) * SMH_ACM_QCAP : nullptr;
    Q.count = 0;
    Q.matches = 0;
    uint32_t cnt = 0;
    uint32_t cur[20], nxt[20];
    auto is_fast = [&](uint64_t kk) { return kk >= 1 && kk < n_chunks && (kk + 1) * chunk_bytes <= C.n; };
    auto load = [&](uint64_t kk, uint32_t (&w)[20]) {
        const uint8_t *p = C.text + smh_uniform64(kk * chunk_bytes) + (uint64_t)lane * SMH_SEG - 16u;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            const smh_u32x4 t = smh_load16(p + 16u * q);
            w[4 * q + 0] = t.v[0];
            w[4 * q + 1] = t.v[1];
            w[4 * q + 2] = t.v[2];
            w[4 * q + 3] = t.v[3];
        }
    };
    uint64_t k = S.take(n_chunks);
    bool cur_fast = is_fast(k) && C.K <= 17;
    if (cur_fast) load(k, cur);
    while (k < n_chunks) {
        const uint64_t kn = S.take(n_chunks);
        const bool nxt_fast = is_fast(kn) && C.K <= 17;
        if (SMH_PREFETCH && nxt_fast) load(kn, nxt);
        const uint64_t a = smh_uniform64(k * chunk_bytes) + (uint64_t)lane * SMH_SEG;
        if (cur_fast) cnt += smh_acm_lane_fast<E, SIGMA>(a, cur, tab, C, Q);
        else cnt += smh_acm_lane_slow<E>(a, tab_g, C);
        if (nxt_fast) {
            if (SMH_PREFETCH) {
#pragma unroll
                for (int q = 0; q < 20; ++q) cur[q] = nxt[q];
            } else {
                load(kn, cur);
            }
        }
        cur_fast = nxt_fast;
        k = kn;
    }
    smh_acm_drain(Q, C);
    return cnt + Q.matches;
}

#endif
