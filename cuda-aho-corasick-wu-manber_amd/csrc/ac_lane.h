/*
 * csrc/ac_lane.h -- what one lane of the Aho-Corasick kernels does.
 *
 * Replaces the per-thread loops of the reference's ac_kernel1..5b
 * (cuda/cuda_ac.cu:23-592).  The quantity computed is the one search_ac
 * returns (ac/ac.c:198-222): the number of text positions at which the
 * automaton is in an accepting state.  With all patterns of length m that is
 * the number of occurrences, and an occurrence is counted by the lane whose
 * segment contains its START, so lanes never double count and never need each
 * other's state (cuda/cuda_ac.cu:31-34 relies on the same argument).
 *
 * Two stages (smh_internal.h "AC" for the table formats):
 *   scan   : every lane walks the depth-K automaton, whole in LDS, over its
 *            64-byte segment plus K-1 halo bytes, one lookup per symbol
 *            (stride 1) or per two symbols (stride 2).  K == m: a flagged
 *            transition is a match.  K < m: it is a candidate -- a K-symbol
 *            pattern prefix ends here -- and is pushed, with its depth-K row,
 *            onto the wave's queue (ballot + prefix count: smh_ac_emit).
 *   verify : when the queue fills, 64 candidates at a time are walked down the
 *            goto edges of the full DFA in HBM for the remaining m-K symbols
 *            (smh_ac_deep_walk).  Candidates are rare by construction (K is
 *            chosen so), so this stage costs a few percent.
 */
#ifndef SMH_AC_LANE_H
#define SMH_AC_LANE_H

#include "lane_common.h"
#include <utility>

template <typename E> struct smh_ac_entry;
template <> struct smh_ac_entry<uint16_t> {
    static constexpr uint32_t FLAG_SHIFT = 15, MASK = 0x7FFFu;
};
template <> struct smh_ac_entry<uint32_t> {
    static constexpr uint32_t FLAG_SHIFT = 31, MASK = 0x7FFFFFFFu;
};

#define SMH_AC_QCAP 256u /* queue entries per wave (HBM workspace, 8 bytes each) */

/* everything the verify stage needs; wave-uniform */
struct smh_ac_verify_ctx {
    const uint8_t *text;
    uint64_t n;
    int m;
    int K;
    int sigma;
    const void *full;            /* full DFA in HBM */
    int full_entry_bytes;
    const uint32_t *depth_first; /* [d] = first row with depth >= d; padded with `rows` */
    const void *trunc1;          /* stride-1 depth-K table in HBM */
    int trunc1_entry_bytes;
};

SMH_LANE uint32_t smh_entry_at(const void *t, int eb, uint64_t i)
{
    return eb == 2 ? (uint32_t)((const uint16_t *)t)[i] : ((const uint32_t *)t)[i];
}

/*
 * A K-symbol pattern prefix ends at text[q] and corresponds to depth-K row `row` of the DFA.
 * Follow goto edges for the remaining m-K symbols: an edge exists iff the next row is one level
 * deeper (rows are numbered breadth-first, so depth(r) >= d  <=>  r >= depth_first[d]); the
 * last edge, into an accepting leaf, is the FLAG bit.  lazy: `row` is the row BEFORE text[q] was
 * consumed (stride-2 scan, candidate on the first symbol of a pair) -- one stride-1 step first.
 */
SMH_LANE uint32_t smh_ac_deep_walk(const smh_ac_verify_ctx &V, uint64_t q, uint32_t row, bool lazy)
{
    if (lazy) {
        uint32_t c0 = V.text[q];
        if (c0 >= (uint32_t)V.sigma) c0 = 0;
        const uint32_t e = smh_entry_at(V.trunc1, V.trunc1_entry_bytes, (uint64_t)row * (uint32_t)V.sigma + c0);
        row = e & (V.trunc1_entry_bytes == 2 ? 0x7FFFu : 0x7FFFFFFFu);
    }
    if (q + (uint64_t)(V.m - V.K) >= V.n) return 0;
    const uint32_t fshift = V.full_entry_bytes == 2 ? 15u : 31u;
    const uint32_t fmask = (1u << fshift) - 1u;
    for (int t = V.K; t < V.m; ++t) {
        const uint32_t c = V.text[q + 1 + (uint64_t)(t - V.K)];
        if (c >= (uint32_t)V.sigma) return 0;
        const uint32_t e = smh_entry_at(V.full, V.full_entry_bytes, (uint64_t)row * (uint32_t)V.sigma + c);
        if (e >> fshift) return 1;
        row = e & fmask;
        if (row < V.depth_first[t + 1]) return 0;
    }
    return 0;
}

/* per-wave candidate queue; `count` and `matches` are per lane on the CPU emulation */
struct smh_ac_queue {
    uint64_t *slots; /* SMH_AC_QCAP entries in HBM, private to this wave */
    uint32_t count;  /* wave-uniform */
    uint32_t matches;
};

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE void smh_ac_drain(smh_ac_queue &Q, const smh_ac_verify_ctx &V)
{
    if (Q.count == 0) return;
    /* the entries were written by this wave with write-through (sc1) stores; wait for them,
     * then read them back past the L1 */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t i = lane; i < Q.count; i += 64u) {
        const uint64_t ent = __hip_atomic_load(Q.slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t rl = (uint32_t)(ent >> 40);
        Q.matches += smh_ac_deep_walk(V, ent & 0xFFFFFFFFFFull, rl & 0x7FFFFFu, (rl >> 23) != 0);
    }
    Q.count = 0;
}

/* wavefront-level compaction: lanes with `cond` append {position, row} to the wave's queue.
 * Must be called in wave-uniform control flow. */
SMH_LANE void smh_ac_emit(smh_ac_queue &Q, const smh_ac_verify_ctx &V, bool cond, uint64_t pos, uint32_t row, bool lazy)
{
    const uint64_t mask = __ballot(cond);
    if (mask == 0) return;
    const uint32_t np = (uint32_t)__popcll(mask);
    if (Q.count + np > SMH_AC_QCAP) smh_ac_drain(Q, V);
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (cond) {
        const uint64_t ent = pos | ((uint64_t)(row | (lazy ? 0x800000u : 0u)) << 40);
        __hip_atomic_store(Q.slots + Q.count + before, ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    Q.count += np;
}
#else
SMH_LANE void smh_ac_drain(smh_ac_queue &, const smh_ac_verify_ctx &) {}
SMH_LANE void smh_ac_emit(smh_ac_queue &Q, const smh_ac_verify_ctx &V, bool cond, uint64_t pos, uint32_t row, bool lazy)
{
    if (cond) Q.matches += smh_ac_deep_walk(V, pos, row, lazy);
}
#endif

/* ------------------------------------------------------------------ scan building blocks */
template <typename E, int SIGMA>
SMH_LANE uint32_t smh_ac_step1(uint32_t &row, uint32_t c, const E *tab, int sigma_rt)
{
    const uint32_t sigma = SIGMA ? (uint32_t)SIGMA : (uint32_t)sigma_rt;
    /* symbols must be < alphabet (as in the reference, which indexes next[] with the raw
     * byte: ac/ac.c:209); an out-of-range byte is folded so it can never index past the table */
    if (SIGMA && (SIGMA & (SIGMA - 1)) == 0)
        c &= (uint32_t)(SIGMA - 1);
    else if (c >= sigma)
        c = 0;
    const uint32_t e = tab[row * sigma + c];
    row = e & smh_ac_entry<E>::MASK;
    return e >> smh_ac_entry<E>::FLAG_SHIFT;
}

/* symbol pair code c1*4+c2 of bytes (2k, 2k+1) of a text word, alphabet 4 */
SMH_LANE uint32_t smh_pair_code(uint32_t word, int k)
{
    const uint32_t h = word >> (16 * k);
    return ((h & 3u) << 2) | ((h >> 8) & 3u);
}

template <typename E, int SIGMA, int STRIDE, int HC, int NCH, bool EXACT> struct smh_ac_scan_ctx {
    const E *tab;   /* LDS: depth-K automaton */
    int sigma_rt;
    int halo;       /* K - 1 */
    const uint32_t *depth_first;
    const uint8_t *text;
    const uint64_t *a;    /* segment offsets of the NCH chains */
    const uint32_t *tail; /* wave-uniform: the bytes that follow the wave-chunk */
    const smh_ac_verify_ctx *V;
    smh_ac_queue *Q;
};

/* one scan step of all NCH chains on bytes taken from `word[j]` at byte index `b` (stride 1) or
 * byte pair (b, b+1) (stride 2); `pos0[j] + b` is the text position of the first byte.
 * second_valid: whether a flag on the second byte of a pair may be used (false on the odd tail of
 * the halo, where that byte already belongs to the next lane's starts). */
template <typename E, int SIGMA, int STRIDE, int HC, int NCH, bool EXACT>
SMH_LANE void smh_ac_scan_step(const smh_ac_scan_ctx<E, SIGMA, STRIDE, HC, NCH, EXACT> &c, const uint32_t (&word)[NCH],
                               int b, const uint64_t (&pos0)[NCH], bool second_valid, uint32_t (&row)[NCH],
                               uint32_t &cnt)
{
    if (STRIDE == 1) {
        uint32_t f[NCH];
        uint32_t anyf = 0;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            f[j] = smh_ac_step1<E, SIGMA>(row[j], smh_byte_of(word[j], b), c.tab, c.sigma_rt);
            anyf |= f[j];
        }
        if (EXACT) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) cnt += f[j];
        } else if (SMH_WAVE_ANY(anyf != 0)) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) smh_ac_emit(*c.Q, *c.V, f[j] != 0, pos0[j] + (uint64_t)b, row[j], false);
        }
    } else {
        uint32_t f[NCH], prev[NCH];
        uint32_t anyf = 0;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            prev[j] = row[j];
            const uint32_t e = c.tab[row[j] * 16u + smh_pair_code(word[j], b >> 1)];
            row[j] = e & 0x3FFFu;
            f[j] = e >> 14;
            if (!second_valid) f[j] &= 1u;
            anyf |= f[j];
        }
        if (EXACT) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) cnt += (f[j] & 1u) + (f[j] >> 1);
        } else if (SMH_WAVE_ANY(anyf != 0)) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                /* first symbol of the pair: the depth-K row is not in the entry -> resolved lazily */
                smh_ac_emit(*c.Q, *c.V, (f[j] & 1u) != 0, pos0[j] + (uint64_t)b, prev[j], true);
                smh_ac_emit(*c.Q, *c.V, (f[j] & 2u) != 0, pos0[j] + (uint64_t)b + 1u, row[j], false);
            }
        }
    }
}

/*
 * One halo step with a COMPILE-TIME byte index H, so the text registers are
 * indexed statically (a runtime-indexed register array would be demoted to
 * scratch memory).  Returns false when the wave is done with the halo.  The
 * steps are chained with a short-circuit fold in smh_ac_halo_all -- hipcc does
 * not unroll a loop whose exit depends on a wave-wide vote.
 *
 * Early exit: after H halo bytes a lane can only still start-own a candidate if its state is at
 * least H+1 deep -- otherwise the longest pattern prefix ending here starts beyond the segment.
 * Rows are numbered breadth-first, so "depth >= H+1" is "row >= depth_first[H+1]".  Lanes that
 * are past that point keep stepping with the rest of the wave (it is harmless: they cannot reach
 * depth K inside the halo), which keeps the step free of divergence.
 */
template <int H, typename E, int SIGMA, int STRIDE, int HC, int NCH, bool EXACT>
SMH_LANE bool smh_ac_halo_step(const smh_ac_scan_ctx<E, SIGMA, STRIDE, HC, NCH, EXACT> &c,
                               const uint32_t (&w)[NCH][16], uint32_t (&hw)[NCH], uint32_t (&row)[NCH],
                               uint32_t &cnt)
{
    if (H % STRIDE != 0) return true; /* stride 2 consumes bytes H and H+1 at even H */
    if (H >= c.halo) return false;
    const uint32_t need = c.depth_first[H + 1];
    bool any = false;
#pragma unroll
    for (int j = 0; j < NCH; ++j) any |= row[j] >= need;
    if (!SMH_WAVE_ANY(any)) return false;
    if ((H & 3) == 0) {
        /* next halo dword: the neighbour lane's segment word H/4 (all lanes active here: every
         * branch above is wave-uniform); lane 63 takes the next chain's lane 0, or the tail */
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const uint32_t edge = j + 1 < NCH ? smh_first_lane(w[j + 1 < NCH ? j + 1 : j][H >> 2]) : c.tail[H >> 2];
            hw[j] = smh_next_lane_word(w[j][H >> 2], edge, c.text, c.a[j] + SMH_SEG + (uint64_t)H);
        }
    }
    uint64_t pos0[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) pos0[j] = c.a[j] + SMH_SEG + (uint64_t)(H & ~3);
    smh_ac_scan_step(c, hw, H & 3, pos0, H + 1 < c.halo, row, cnt);
    return true;
}

template <typename E, int SIGMA, int STRIDE, int HC, int NCH, bool EXACT, int... Hs>
SMH_LANE void smh_ac_halo_all(const smh_ac_scan_ctx<E, SIGMA, STRIDE, HC, NCH, EXACT> &c, const uint32_t (&w)[NCH][16],
                              uint32_t (&row)[NCH], uint32_t &cnt, std::integer_sequence<int, Hs...>)
{
    uint32_t hw[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) hw[j] = 0;
    (void)(smh_ac_halo_step<Hs>(c, w, hw, row, cnt) && ...);
}

/*
 * Fast path: NCH segments per lane, each fully inside the text together with
 * 16*HC bytes after it (the caller guarantees a[j] + 64 + 16*HC <= n and
 * 16*HC >= K-1).  Only the 64 segment bytes are loaded; the halo bytes come out
 * of the neighbouring lane's registers (smh_next_lane_word).  The NCH automata
 * are independent dependency chains stepped in lock-step, so the LDS latency of
 * one hides behind the others.
 */
template <int NCH>
SMH_LANE void smh_ac_load_segments(const uint8_t *text, const uint64_t (&a)[NCH], uint32_t (&w)[NCH][16])
{
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const smh_u32x4 t = smh_load16(text + a[j] + 16u * q);
            w[j][4 * q + 0] = t.v[0];
            w[j][4 * q + 1] = t.v[1];
            w[j][4 * q + 2] = t.v[2];
            w[j][4 * q + 3] = t.v[3];
        }
}

template <typename E, int SIGMA, int STRIDE, int HC, int NCH, bool EXACT>
SMH_LANE uint32_t smh_ac_lane_fast(const uint8_t *text, const uint64_t (&a)[NCH], const uint32_t (&w)[NCH][16],
                                   const uint32_t *tail, const E *tab, int sigma_rt, int K,
                                   const uint32_t *depth_first, const smh_ac_verify_ctx &V, smh_ac_queue &Q)
{
    uint32_t row[NCH], cnt = 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) row[j] = 0;
    smh_ac_scan_ctx<E, SIGMA, STRIDE, HC, NCH, EXACT> ctx{tab, sigma_rt, K - 1, depth_first, text, a, tail, &V, &Q};

#pragma unroll
    for (int i = 0; i < 64; i += STRIDE) {
        uint32_t word[NCH];
        uint64_t pos0[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            word[j] = w[j][i >> 2];
            pos0[j] = a[j] + (uint64_t)(i & ~3);
        }
        smh_ac_scan_step(ctx, word, i & 3, pos0, true, row, cnt);
    }
    smh_ac_halo_all(ctx, w, row, cnt, std::make_integer_sequence<int, 16 * HC>{});
    return cnt;
}

/* Slow path: any segment, byte loads with bounds checks, stride-1 depth-K table from HBM,
 * candidates verified on the spot.  Used for the last wave-chunk(s) of a text and for texts
 * shorter than one wave-chunk. */
SMH_LANE uint32_t smh_ac_lane_slow(const smh_ac_verify_ctx &V, uint64_t n_starts, uint64_t a)
{
    if (a >= n_starts) return 0;
    uint64_t own_end = a + SMH_SEG;
    if (own_end > n_starts) own_end = n_starts;
    /* K-symbol prefixes that START in [a, own_end) END before own_end + K - 1 */
    uint64_t stop = own_end + (uint64_t)(V.K - 1);
    if (stop > V.n) stop = V.n;
    const uint32_t tmask = V.trunc1_entry_bytes == 2 ? 0x7FFFu : 0x7FFFFFFFu;
    const uint32_t tshift = V.trunc1_entry_bytes == 2 ? 15u : 31u;
    uint32_t row = 0, cnt = 0;
    for (uint64_t i = a; i < stop; ++i) {
        uint32_t c = V.text[i];
        if (c >= (uint32_t)V.sigma) c = 0;
        const uint32_t e = smh_entry_at(V.trunc1, V.trunc1_entry_bytes, (uint64_t)row * (uint32_t)V.sigma + c);
        row = e & tmask;
        if (e >> tshift) cnt += V.K >= V.m ? 1u : smh_ac_deep_walk(V, i, row, false);
    }
    return cnt;
}

/*
 * SMH_VARIANT_TABLE: the reference-layout goto / supply / final tables walked
 * as they are (cuda/cuda_ac.cu:584-591): -1 = no edge, follow supply links.
 * Lane owns the starts [a, a + span).
 */
SMH_LANE uint32_t smh_ac_lane_table(const uint8_t *text, uint64_t n, uint64_t n_starts, uint64_t a,
                                    uint64_t span, const int32_t *transition, const uint32_t *supply,
                                    const uint32_t *final, int alphabet, int m)
{
    if (a >= n_starts) return 0;
    uint64_t own_end = a + span;
    if (own_end > n_starts) own_end = n_starts;
    uint64_t stop = own_end + (uint64_t)(m - 1);
    if (stop > n) stop = n;
    uint32_t cnt = 0;
    int32_t r = 0, s;
    for (uint64_t i = a; i < stop; ++i) {
        uint32_t c = text[i];
        if (c >= (uint32_t)alphabet) c = 0; /* out-of-range byte: see smh_ac_step1 */
        while ((s = transition[(uint64_t)r * (uint32_t)alphabet + c]) == -1) r = (int32_t)supply[r];
        r = s;
        cnt += final[r];
    }
    return cnt;
}

/*
 * Whole-grid work distribution for one lane (thread `gthread` of `nthreads`,
 * 64 lanes per wave): wave-chunks of NCH*4 KiB are dealt round-robin to waves,
 * so at any moment the resident waves stream one contiguous window of text.
 */
template <typename E, int SIGMA, int STRIDE, int HC, int NCH, bool EXACT>
SMH_LANE uint32_t smh_ac_thread(uint64_t gthread, uint64_t nthreads, const E *tab, const smh_ac_verify_ctx &V,
                                uint64_t *queue_base)
{
    if (V.n < (uint64_t)V.m) return 0;
    const uint64_t n_starts = V.n - (uint64_t)V.m + 1;
    const uint64_t chunk_bytes = (uint64_t)SMH_SEG * 64u * NCH;
    const uint64_t n_chunks = (n_starts + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    const uint64_t wave = gthread >> 6, nwaves = nthreads >> 6;
    smh_ac_queue Q;
    Q.slots = queue_base ? queue_base + smh_uniform64(wave) * SMH_AC_QCAP : nullptr;
    Q.count = 0;
    Q.matches = 0;
    uint32_t cnt = 0;
    /* software pipeline: the segments of the wave's NEXT chunk are requested before the current
     * chunk is scanned, so the HBM latency of a chunk hides behind a whole chunk of lookups */
    uint32_t cur[NCH][16], nxt[NCH][16];
    uint64_t k = wave;
    bool cur_fast = false;
    if (k < n_chunks) {
        const uint64_t base = smh_uniform64(k * chunk_bytes); /* same for the 64 lanes of a wave */
        cur_fast = base + chunk_bytes + 16u * HC <= V.n;
        if (cur_fast) {
            uint64_t a[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) a[j] = base + ((uint64_t)j * 64u + lane) * SMH_SEG;
            smh_ac_load_segments<NCH>(V.text, a, cur);
        }
    }
    while (k < n_chunks) {
        const uint64_t base = smh_uniform64(k * chunk_bytes);
        const uint64_t kn = k + nwaves;
        const uint64_t base_n = smh_uniform64(kn * chunk_bytes);
        const bool nxt_fast = kn < n_chunks && base_n + chunk_bytes + 16u * HC <= V.n;
        if (nxt_fast) {
            uint64_t an[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) an[j] = base_n + ((uint64_t)j * 64u + lane) * SMH_SEG;
            smh_ac_load_segments<NCH>(V.text, an, nxt);
        }
        if (cur_fast) {
            uint64_t a[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) a[j] = base + ((uint64_t)j * 64u + lane) * SMH_SEG;
            const uint32_t *tail = reinterpret_cast<const uint32_t *>(V.text + base + chunk_bytes);
            cnt += smh_ac_lane_fast<E, SIGMA, STRIDE, HC, NCH, EXACT>(V.text, a, cur, tail, tab, V.sigma, V.K,
                                                                      V.depth_first, V, Q);
        } else {
            for (int j = 0; j < NCH; ++j)
                cnt += smh_ac_lane_slow(V, n_starts, base + ((uint64_t)j * 64u + lane) * SMH_SEG);
        }
        if (nxt_fast) {
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) cur[j][q] = nxt[j][q];
        }
        cur_fast = nxt_fast;
        k = kn;
    }
    if (!EXACT) smh_ac_drain(Q, V);
    return cnt + Q.matches;
}

#define SMH_AC_TABLE_SPAN 256u /* starts per lane in the table-walking kernel */
SMH_LANE uint32_t smh_ac_table_thread(uint64_t gthread, uint64_t nthreads, const uint8_t *text, uint64_t n,
                                      int m, const int32_t *transition, const uint32_t *supply,
                                      const uint32_t *final, int alphabet)
{
    if (n < (uint64_t)m) return 0;
    const uint64_t n_starts = n - (uint64_t)m + 1;
    uint32_t cnt = 0;
    for (uint64_t a = gthread * SMH_AC_TABLE_SPAN; a < n_starts; a += nthreads * SMH_AC_TABLE_SPAN)
        cnt += smh_ac_lane_table(text, n, n_starts, a, SMH_AC_TABLE_SPAN, transition, supply, final,
                                 alphabet, m);
    return cnt;
}

#endif
