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
 * DFA entry: next row | FLAG, FLAG = top bit = "accepting after this step".
 * Rows [0, hot_rows) are in LDS (`hot`), every row is also in HBM (`full`).
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

/* one automaton step; returns the accepting flag (0/1) */
template <typename E, int SIGMA, bool ALLHOT>
SMH_LANE uint32_t smh_ac_step(uint32_t &row, uint32_t c, const E *hot, const E *full,
                              uint32_t hot_rows, int sigma_rt)
{
    const uint32_t sigma = SIGMA ? (uint32_t)SIGMA : (uint32_t)sigma_rt;
    /* symbols must be < alphabet (as in the reference, which indexes next[] with the raw
     * byte: ac/ac.c:209); an out-of-range byte is folded so it can never index past the table */
    if (SIGMA && (SIGMA & (SIGMA - 1)) == 0)
        c &= (uint32_t)(SIGMA - 1);
    else if (c >= sigma)
        c = 0;
    const uint32_t idx = row * sigma + c;
    uint32_t e;
    if (ALLHOT || row < hot_rows)
        e = hot[idx];
    else
        e = full[idx];
    row = e & smh_ac_entry<E>::MASK;
    return e >> smh_ac_entry<E>::FLAG_SHIFT;
}

/*
 * One halo step with a COMPILE-TIME byte index H, so the text registers are
 * indexed statically (a runtime-indexed register array would be demoted to
 * scratch memory).  Returns false when the wave is done with the halo.  The
 * steps are chained with a short-circuit fold in smh_ac_halo_all -- hipcc does
 * not unroll a loop whose exit depends on a wave-wide vote.
 */
template <typename E, int SIGMA, int HC, int NCH, bool ALLHOT> struct smh_ac_halo_ctx {
    const E *hot;
    const E *full;
    uint32_t hot_rows;
    int sigma_rt;
    int halo;
    const uint32_t *depth_first;
};

template <int H, typename E, int SIGMA, int HC, int NCH, bool ALLHOT>
SMH_LANE bool smh_ac_halo_step(const smh_ac_halo_ctx<E, SIGMA, HC, NCH, ALLHOT> &c,
                               const uint32_t (&w)[NCH][16 + 4 * HC], uint32_t (&row)[NCH], uint32_t &cnt)
{
    if (H >= c.halo) return false;
    const uint32_t need = c.depth_first[H + 1];
    bool any = false;
    bool act[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        act[j] = row[j] >= need;
        any |= act[j];
    }
    if (!SMH_WAVE_ANY(any)) return false;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
        if (act[j])
            cnt += smh_ac_step<E, SIGMA, ALLHOT>(row[j], smh_byte_of(w[j][16 + (H >> 2)], H & 3), c.hot, c.full,
                                                 c.hot_rows, c.sigma_rt);
    return true;
}

template <typename E, int SIGMA, int HC, int NCH, bool ALLHOT, int... Hs>
SMH_LANE void smh_ac_halo_all(const smh_ac_halo_ctx<E, SIGMA, HC, NCH, ALLHOT> &c,
                              const uint32_t (&w)[NCH][16 + 4 * HC], uint32_t (&row)[NCH], uint32_t &cnt,
                              std::integer_sequence<int, Hs...>)
{
    (void)(smh_ac_halo_step<Hs>(c, w, row, cnt) && ...);
}

/*
 * Fast path: NCH segments per lane, each fully inside the text together with
 * its 16*HC-byte post-halo (the caller guarantees a[j] + 64 + 16*HC <= n and
 * 16*HC >= m-1).  The NCH automata are independent dependency chains and are
 * stepped in lock-step so the LDS latency of one hides behind the others.
 *
 * Halo early exit: after h halo bytes a lane can stop as soon as its state is
 * shallower than h+1 -- the longest pattern prefix ending here then starts
 * beyond the segment, so every later match belongs to the next lane.  Rows are
 * numbered breadth-first, so "depth >= h+1" is "row >= depth_first[h+1]".
 */
template <typename E, int SIGMA, int HC, int NCH, bool ALLHOT>
SMH_LANE uint32_t smh_ac_lane_fast(const uint8_t *text, const uint64_t (&a)[NCH], const E *hot,
                                   const E *full, uint32_t hot_rows, int sigma_rt, int m,
                                   const uint32_t *depth_first)
{
    uint32_t w[NCH][16 + 4 * HC];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int q = 0; q < 4 + HC; ++q) {
            const smh_u32x4 t = smh_load16(text + a[j] + 16u * q);
            w[j][4 * q + 0] = t.v[0];
            w[j][4 * q + 1] = t.v[1];
            w[j][4 * q + 2] = t.v[2];
            w[j][4 * q + 3] = t.v[3];
        }
    uint32_t row[NCH], cnt = 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) row[j] = 0;

#pragma unroll
    for (int i = 0; i < 64; ++i)
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            cnt += smh_ac_step<E, SIGMA, ALLHOT>(row[j], smh_byte_of(w[j][i >> 2], i & 3), hot, full,
                                                 hot_rows, sigma_rt);

    smh_ac_halo_ctx<E, SIGMA, HC, NCH, ALLHOT> ctx{hot, full, hot_rows, sigma_rt, m - 1, depth_first};
    smh_ac_halo_all(ctx, w, row, cnt, std::make_integer_sequence<int, 16 * HC>{});
    return cnt;
}

/* Slow path: any segment, byte loads with bounds checks.  Used for the last
 * wave-chunk(s) of a text and for texts shorter than one wave-chunk. */
template <typename E, int SIGMA, bool ALLHOT>
SMH_LANE uint32_t smh_ac_lane_slow(const uint8_t *text, uint64_t n, uint64_t n_starts, uint64_t a,
                                   const E *hot, const E *full, uint32_t hot_rows, int sigma_rt, int m)
{
    if (a >= n_starts) return 0;
    uint64_t own_end = a + SMH_SEG;
    if (own_end > n_starts) own_end = n_starts;
    uint64_t stop = own_end + (uint64_t)(m - 1); /* <= n because own_end <= n - m + 1 */
    if (stop > n) stop = n;
    uint32_t row = 0, cnt = 0;
    for (uint64_t i = a; i < stop; ++i)
        cnt += smh_ac_step<E, SIGMA, ALLHOT>(row, text[i], hot, full, hot_rows, sigma_rt);
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
        if (c >= (uint32_t)alphabet) c = 0; /* out-of-range byte: see smh_ac_step */
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
 * HC == 0 means "no fast path" (m - 1 > 64): every segment takes the slow path.
 */
template <typename E, int SIGMA, int HC, int NCH, bool ALLHOT>
SMH_LANE uint32_t smh_ac_thread(uint64_t gthread, uint64_t nthreads, const uint8_t *text, uint64_t n,
                                int m, const E *hot, const E *full, uint32_t hot_rows, int sigma_rt,
                                const uint32_t *depth_first)
{
    if (n < (uint64_t)m) return 0;
    const uint64_t n_starts = n - (uint64_t)m + 1;
    const uint64_t chunk_bytes = (uint64_t)SMH_SEG * 64u * NCH;
    const uint64_t n_chunks = (n_starts + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    const uint64_t wave = gthread >> 6, nwaves = nthreads >> 6;
    uint32_t cnt = 0;
    for (uint64_t k = wave; k < n_chunks; k += nwaves) {
        const uint64_t base = k * chunk_bytes;
        if (HC > 0 && base + chunk_bytes + 16u * HC <= n) {
            uint64_t a[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) a[j] = base + ((uint64_t)j * 64u + lane) * SMH_SEG;
            cnt += smh_ac_lane_fast<E, SIGMA, (HC > 0 ? HC : 1), NCH, ALLHOT>(text, a, hot, full, hot_rows,
                                                                            sigma_rt, m, depth_first);
        } else {
            for (int j = 0; j < NCH; ++j)
                cnt += smh_ac_lane_slow<E, SIGMA, ALLHOT>(text, n, n_starts,
                                                          base + ((uint64_t)j * 64u + lane) * SMH_SEG, hot,
                                                          full, hot_rows, sigma_rt, m);
        }
    }
    return cnt;
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
