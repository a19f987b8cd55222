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
#include "wm_lane.h" /* smh_wm_verify: the verify stage hashes the window when the handle carries a verify table */
#include <utility>

template <typename E> struct smh_ac_entry;
template <> struct smh_ac_entry<uint16_t> {
    static constexpr uint32_t FLAG_SHIFT = 15, MASK = 0x7FFFu;
};
template <> struct smh_ac_entry<uint32_t> {
    static constexpr uint32_t FLAG_SHIFT = 31, MASK = 0x7FFFFFFFu;
};

#define SMH_AC_QCAP 256u /* queue entries per wave (HBM workspace, 8 bytes each) */

/* What only the RARE paths read -- the verify stage (a candidate of a depth-cut plan), the bounds-checked walk over the
 * text's last piece, the per-segment positions kernel: the full DFA, the stride-1 depth-K table, the hash-verify tables.
 * It lives in device memory (one copy per handle and device, uploaded with the table set) and the kernels receive a
 * POINTER to it: passed by value these eleven fields were 20 scalar registers that the compiler loaded in the kernel's
 * prologue and kept alive -- or spilled to VGPR lanes and read back with v_readlane inside the halo steps -- across the
 * whole scan loop; the depth-cut kernels carried four times the scalar spill code of the exact ones and ran 0.02-0.03
 * ms/GiB behind them (round 3: the same kernel with the verify stage compiled out, 0.200 -> 0.182). */
struct smh_ac_cold_ctx {
    const void *full;            /* full DFA in HBM */
    const uint32_t *depth_first; /* [d] = first row with depth >= d; padded with `rows` */
    const void *trunc1;          /* stride-1 depth-K table in HBM */
    /* hash verify (ac_host.c hv_wm): the Wu-Manber verify table and the zero-padded patterns, or NULL: walk the DFA */
    const uint32_t *hv_verify;
    const uint8_t *hv_pats;
    int full_entry_bytes;
    int trunc1_entry_bytes;
    int hv_log2;
    int reserved;
};

/* what a scan needs from its caller; wave-uniform */
struct smh_ac_verify_ctx {
    const uint8_t *text;
    uint64_t n;
    int m;
    int K;
    int sigma;
    const smh_ac_cold_ctx *cold; /* device memory (the emulator: host memory) */
    smh_pos_out pos;             /* positions mode: where match END columns go (cursor == NULL: counting) */
};

/* depth_first[0..71] BY VALUE: as a kernel argument it is read with scalar loads from the kernarg
 * segment.  Read through a pointer it becomes a vector load as soon as the kernel also stores to
 * global memory (the candidate queue), and the s_waitcnt vmcnt(0) in front of its use would then
 * also wait for the prefetched next chunk -- measured as a 30 % slowdown. */
#define SMH_AC_DF_LEN 72
struct smh_ac_df {
    uint32_t v[SMH_AC_DF_LEN];
};

SMH_LANE uint32_t smh_entry_at(const void *t, int eb, uint64_t i)
{
    return eb == 2 ? (uint32_t)((const uint16_t *)t)[i] : ((const uint32_t *)t)[i];
}

/*
 * A K-symbol pattern prefix ends at text[q].  Follow goto edges of the full DFA for the remaining
 * symbols: an edge exists iff the next row is one level deeper (rows are numbered breadth-first,
 * so depth(r) >= d  <=>  r >= depth_first[d]); the last edge, into an accepting leaf, is the FLAG
 * bit.  What the scan knew about the candidate decides where the walk starts:
 *   SMH_CAND_ROW   `row` is the depth-K row reached at text[q]: walk symbols K .. m-1
 *   SMH_CAND_LAZY  `row` is the row BEFORE text[q] was consumed (stride-2 scan, candidate on the
 *                  first symbol of a pair): one stride-1 step of the depth-K table first
 *   SMH_CAND_ROOT  only the position is known (stride-2 scan that records candidates as bits):
 *                  walk all m symbols from the root, starting at q - K + 1
 */
#define SMH_CAND_ROW 0u
#define SMH_CAND_LAZY 1u
#define SMH_CAND_ROOT 2u

SMH_LANE uint32_t smh_ac_deep_walk(const smh_ac_verify_ctx &V, uint64_t q, uint32_t row, uint32_t kind)
{
    const smh_ac_cold_ctx &C = *V.cold;
    int t0 = V.K;
    uint64_t start = q + 1 - (uint64_t)V.K; /* text position of the pattern's first symbol */
    if (C.hv_verify && (kind == SMH_CAND_ROOT || V.m - V.K > 3)) {
        /* three dependent loads (window, bucket, pattern) instead of one per remaining symbol */
        if (start + (uint64_t)V.m > V.n) return 0;
        smh_wm_params P = {};
        P.m = V.m;
        P.verify_log2 = C.hv_log2;
        P.verify = C.hv_verify;
        P.pat_sorted = C.hv_pats;
        return smh_wm_verify(V.text, start + (uint64_t)V.m - 1u, P);
    }
    if (kind == SMH_CAND_LAZY) {
        uint32_t c0 = V.text[q];
        if (c0 >= (uint32_t)V.sigma) c0 = 0;
        const uint32_t e = smh_entry_at(C.trunc1, C.trunc1_entry_bytes, (uint64_t)row * (uint32_t)V.sigma + c0);
        row = e & (C.trunc1_entry_bytes == 2 ? 0x7FFFu : 0x7FFFFFFFu);
    } else if (kind == SMH_CAND_ROOT) {
        row = 0;
        t0 = 0;
    }
    if (start + (uint64_t)V.m > V.n) return 0;
    const uint32_t fshift = C.full_entry_bytes == 2 ? 15u : 31u;
    const uint32_t fmask = (1u << fshift) - 1u;
    /* one DEPENDENT load per step (the DFA entry); the next text byte and the depth bound do not depend on it and are
     * requested beside it -- three loads in sequence per step made the walk that ends every wave of a depth-cut plan
     * 20 us long (tools/wavetrace.py: median wave 193 us against 173 us for the same image with K = m) */
    uint32_t c = V.text[start + (uint64_t)t0];
    for (int t = t0; t < V.m; ++t) {
        const uint32_t need = C.depth_first[t + 1];
        const uint32_t cn = t + 1 < V.m ? V.text[start + (uint64_t)t + 1u] : 0u;
        if (c >= (uint32_t)V.sigma) return 0;
        const uint32_t e = smh_entry_at(C.full, C.full_entry_bytes, (uint64_t)row * (uint32_t)V.sigma + c);
        if (e >> fshift) return 1;
        row = e & fmask;
        if (row < need) return 0;
        c = cn;
    }
    return 0;
}

/* per-wave candidate queue; `count` and `matches` are per lane on the CPU emulation */
struct smh_ac_queue {
    uint64_t *slots; /* SMH_AC_QCAP entries in HBM, private to this wave */
    uint32_t count;  /* wave-uniform */
    uint32_t matches;
    uint32_t events; /* per lane: candidates this lane queued (smh_stats.h) */
};

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE void smh_ac_drain(smh_ac_queue &Q, const smh_ac_verify_ctx &V)
{
    if (Q.count == 0) return;
    /* the entries were written by this wave with write-through (sc1) stores; wait for them, then read them back past the
     * L1.  (Kept inline: an out-of-line drain measured 17 % slower on the m = 32 set in round 1 and twice as slow in
     * round 3 -- a call anywhere in the kernel puts the text registers on the stack.) */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t lane = threadIdx.x & 63u;
    if (V.pos.cursor) {
        /* positions mode: same walk, and the verified candidates append their match END column
         * (the K-symbol prefix ends at the queued position, the pattern m - K symbols later) */
        for (uint32_t base = 0; base < Q.count; base += 64u) {
            const uint32_t i = base + lane;
            uint64_t hit = 0, q = 0;
            if (i < Q.count) {
                const uint64_t ent = __hip_atomic_load(Q.slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t rl = (uint32_t)(ent >> 40);
                q = ent & 0xFFFFFFFFFFull;
                hit = smh_ac_deep_walk(V, q, rl & 0x3FFFFFu, rl >> 22);
            }
            Q.matches += smh_append_bits(hit, q + (uint64_t)(V.m - V.K), V.pos);
        }
        Q.count = 0;
        return;
    }
    for (uint32_t i = lane; i < Q.count; i += 64u) {
        const uint64_t ent = __hip_atomic_load(Q.slots + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t rl = (uint32_t)(ent >> 40);
        Q.matches += smh_ac_deep_walk(V, ent & 0xFFFFFFFFFFull, rl & 0x3FFFFFu, rl >> 22);
    }
    Q.count = 0;
}

/* the same drain as a real function: the emit sites of the vote-and-replay plans (one per halo step and replayed
 * lookup, hundreds per kernel) call it instead of inlining the walk -- their code size and compile time, not their
 * speed, are what matters (a full queue is rare); the one emit site of the bit-recording plans keeps the inline form */
__device__ __noinline__ static void smh_ac_drain_call(smh_ac_queue *Q, const smh_ac_verify_ctx *V) { smh_ac_drain(*Q, *V); }

/* wavefront-level compaction: lanes with `cond` append {position, row} to the wave's queue.
 * Must be called in wave-uniform control flow. */
template <bool INLINE_DRAIN = true>
SMH_LANE void smh_ac_emit(smh_ac_queue &Q, const smh_ac_verify_ctx &V, bool cond, uint64_t pos, uint32_t row, uint32_t kind)
{
    const uint64_t mask = __ballot(cond);
    if (mask == 0) return;
    const uint32_t np = (uint32_t)__popcll(mask);
    if (Q.count + np > SMH_AC_QCAP) {
        if constexpr (INLINE_DRAIN) smh_ac_drain(Q, V);
        else smh_ac_drain_call(&Q, &V);
    }
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (cond) {
        /* (Round 3 measured what this one store costs the scan: 5-7 % -- 0.200 ms/GiB with it, vector or atomic, 0.188
         * with the very same code and the store left out; on gfx9 vector stores share the vmcnt counter with the loads.
         * Scalar stores (s_store_dwordx2, tools/sstore_probe.hip) would avoid that but their write-back is not ordered by
         * any counter the reader can wait on; not used.) */
        const uint64_t ent = pos | ((uint64_t)(row | (kind << 22)) << 40);
        __hip_atomic_store(Q.slots + Q.count + before, ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    Q.count += np;
    Q.events += cond ? 1u : 0u;
}
#else
SMH_LANE void smh_ac_drain(smh_ac_queue &, const smh_ac_verify_ctx &) {}
template <bool INLINE_DRAIN = true>
SMH_LANE void smh_ac_emit(smh_ac_queue &Q, const smh_ac_verify_ctx &V, bool cond, uint64_t pos, uint32_t row, uint32_t kind)
{
    if (!cond) return;
    const uint32_t hit = smh_ac_deep_walk(V, pos, row, kind);
    Q.matches += hit;
    if (hit && V.pos.cursor) smh_append_bits(1u, pos + (uint64_t)(V.m - V.K), V.pos);
}
#endif

/* ------------------------------------------------------------------ scan table formats
 * A format says how a lane turns (state, text dword) into the next LDS address with as few VALU
 * ops as possible.  The lane state is the raw table ENTRY (next row + flag bits), not the row.
 *   prep(w)        once per text dword
 *   next(e, x, k)  entry after consuming byte k (stride 1) or byte pair k (stride 2) of the dword
 *   flags(e)       stride 1: bit 0; stride 2: bit 0 = after the first symbol, bit 1 = after the second
 *   row(e)         row id (for the early-exit test and for queue entries)
 *   any(e)         nonzero iff a flag is set (used on an OR of many entries)
 */
template <typename E, int SIGMA> struct smh_fmt_s1 { /* stride 1: entry = row | FLAG (top bit) */
    static constexpr int STRIDE = 1;
    static constexpr bool SPARSE = false;
    static constexpr uint32_t FSHIFT = smh_ac_entry<E>::FLAG_SHIFT, MASK = smh_ac_entry<E>::MASK;
    int sigma_rt;
    SMH_MEMBER uint32_t prep(uint32_t w) const { return SIGMA == 4 ? w << (sizeof(E) == 2 ? 1 : 2) : w; }
    SMH_MEMBER uint32_t next(uint32_t e, uint32_t x, int k, const void *tab) const
    {
        if (SIGMA == 4) {
            /* x = w << log2(sizeof(E)): the symbol's byte offset inside the row is a bit field of x */
            const uint32_t c = smh_bfe(x, 8 * k, sizeof(E) == 2 ? 3 : 4);
            const uint32_t addr = ((e & MASK) << (sizeof(E) == 2 ? 3 : 4)) | c;
            return sizeof(E) == 2 ? smh_lds_u16(tab, addr) : smh_lds_u32(tab, addr);
        } else {
            const uint32_t sigma = SIGMA ? (uint32_t)SIGMA : (uint32_t)sigma_rt;
            /* symbols must be < alphabet (as in the reference, which indexes next[] with the raw
             * byte: ac/ac.c:209); an out-of-range byte is folded so it cannot index past the table */
            uint32_t c = smh_byte_of(x, k);
            if (c >= sigma) c = 0;
            const uint32_t addr = ((e & MASK) * sigma + c) * (uint32_t)sizeof(E);
            return sizeof(E) == 2 ? smh_lds_u16(tab, addr) : smh_lds_u32(tab, addr);
        }
    }
    SMH_MEMBER uint32_t flags(uint32_t e) const { return e >> FSHIFT; }
    SMH_MEMBER uint32_t row(uint32_t e) const { return e & MASK; }
    SMH_MEMBER uint32_t any(uint32_t e) const { return e >> FSHIFT; }
};

struct smh_fmt_s2 { /* stride 2, alphabet 4: entry = row | F1 << 14 | F2 << 15, 16 entries per row */
    static constexpr int STRIDE = 2;
    static constexpr bool SPARSE = false;
    /* one v_lshl_or per text dword: the pair codes c1*4+c2 land at bits 8..11 (bytes 0,1) and 24..27
     * (bytes 2,3) with a zero bit below each, so one v_bfe yields the code * 2 = the byte offset in a row */
    SMH_MEMBER uint32_t prep(uint32_t w) const { return (w << 10) | w; }
    SMH_MEMBER uint32_t next(uint32_t e, uint32_t x, int k, const void *tab) const
    {
        const uint32_t c = smh_bfe(x, k == 0 ? 7 : 23, 5);
        return smh_lds_u16(tab, ((e & 0x3FFFu) << 5) | c);
    }
    SMH_MEMBER uint32_t flags(uint32_t e) const { return e >> 14; }
    SMH_MEMBER uint32_t row(uint32_t e) const { return e & 0x3FFFu; }
    SMH_MEMBER uint32_t any(uint32_t e) const { return e >> 14; }
};

/*
 * Hybrid stride 2, alphabet 4 (image built by hyb_build in ac_host.c, which documents the layout):
 * rows below `nf` are full -- 16 two-symbol entries holding the plain next row, never a flag --
 * and the others are lists of 4-byte items.  A step first resolves the lanes that sit in a compact
 * row (rare: the wave votes), walking the row's items until one ends the step or the supply link
 * leads into a full row, and then does the ordinary two-symbol lookup for the lanes still open.
 * Flags are raised only inside that resolution, so the common path carries no flag arithmetic:
 * next_f hands them to a callback.  The lane state is the row id (16 bits).
 */
template <bool CLAMP> struct smh_fmt_s2h_t {
    static constexpr int STRIDE = 2;
    static constexpr bool SPARSE = true;
    /* Full rows have the ids [0, full_rows); item slot s of the compact part has the id SMH_HYB_COMPACT0 + s.  A lane that
     * sits in a compact row issues the common path's full-row lookup like every other lane, and the value is replaced by
     * the resolution below.  Two forms of that lookup's address:
     *   CLAMP = false  (row << 5) | c as it is: for a compact id that is >= 1 MiB, beyond the workgroup's LDS, where a
     *                  ds_read returns 0 and raises nothing -- the architected behaviour of an out-of-range LDS read
     *                  on GCN / CDNA, which the library nevertheless CHECKS once per device before it relies on it
     *                  (smh_lds_oob_reads_zero in ac_kernels.inc: a probe kernel reads the very addresses this path
     *                  can produce).  The common step is v_bfe, v_lshl_or, ds_read_u16 per chain.
     *   CLAMP = true   the id is clamped to row 0 first (one v_cndmask more per step and chain: +14 % VALU work):
     *                  what runs on a device whose probe did not read zeros, and in the one-chain instantiations.
     * The CPU emulation executes the SAME addressing and models the out-of-range read as 0 (`limit` = image bytes). */
    uint32_t nf;    /* ids >= nf are compact: SMH_HYB_COMPACT0 */
    uint32_t cbase; /* byte address of the item slot with id i is i * 4 + cbase (mod 2^32) */
    uint32_t limit; /* bytes of the image in LDS (emulation of the out-of-range read only) */
    static SMH_MEMBER smh_fmt_s2h_t make(uint32_t full_rows, uint32_t image_bytes)
    {
        return smh_fmt_s2h_t{SMH_HYB_COMPACT0, full_rows * 32u - 4u * SMH_HYB_COMPACT0, image_bytes};
    }
    SMH_MEMBER uint32_t full_addr(uint32_t row, uint32_t c) const
    {
        if (CLAMP) return ((row < nf ? row : 0u) << 5) | c;
        return (row << 5) | c;
    }
    /* the full-row lookup: on the GPU a plain ds_read_u16 (out of range: 0); emulated with that rule spelled out */
    SMH_MEMBER uint32_t read_full(const void *tab, uint32_t addr) const
    {
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
        return smh_lds_u16(tab, addr);
#else
        return addr + 2u <= limit ? smh_lds_u16(tab, addr) : 0u;
#endif
    }
    SMH_MEMBER uint32_t prep(uint32_t w) const { return (w << 10) | w; } /* as smh_fmt_s2 */
    /* the lanes that sit in a compact row: `t` is the full-row lookup already issued for the clamped row */
    template <typename F>
    SMH_MEMBER uint32_t resolve(uint32_t row, uint32_t c, uint32_t t, const void *tab, F &&on_flags) const
    {
        bool deep = row >= nf;
        if (SMH_UNLIKELY(SMH_WAVE_ANY(deep))) {
            const uint32_t code = c >> 1;
            uint32_t r = row, acc = 0;
            bool done = false, moved = false;
            do {
                if (deep) {
                    const uint32_t addr = r * 4u + cbase;
                    const uint32_t rec = smh_lds_u32(tab, addr);
                    const bool hit = ((code ^ smh_bfe(rec, 16, 4)) & smh_bfe(rec, 20, 4)) == 0;
                    uint32_t nr = rec & 0xFFFFu;
                    if (hit) {
                        acc |= smh_bfe(rec, 24, 2);
                        if (rec & (1u << 26)) nr = r + 1u;
                        if (rec & (1u << 27)) nr = smh_lds_u32(tab, addr + 4u) & 0xFFFFu;
                        done = (rec & (3u << 26)) != 0;
                    }
                    r = nr;
                    moved = true;
                }
                deep = !done && r >= nf;
            } while (SMH_WAVE_ANY(deep));
            on_flags(acc);
            /* lanes that left their compact row through a supply link now stand in a full row */
            if (moved && !done) t = smh_lds_u16(tab, (r << 5) | c);
            if (done) t = r;
        }
        return t;
    }
    template <typename F>
    SMH_MEMBER uint32_t next_f(uint32_t row, uint32_t x, int k, const void *tab, F &&on_flags) const
    {
        const uint32_t c = smh_bfe(x, k == 0 ? 7 : 23, 5); /* pair code * 2 */
        /* the full-row lookup is issued for every lane before the vote (harmless for a compact row id: see above),
         * so the common path is the plain stride-2 one: shift-or, read */
        const uint32_t t = read_full(tab, full_addr(row, c));
        return resolve(row, c, t, tab, on_flags);
    }
    /* the same step for the N chains of a lane with ONE vote in the common path (the deepest of the N rows decides;
     * per chain it was a compare, a mask move and a branch each: 173 scalar instructions per 4 KiB of text against 39
     * in the plain stride-2 kernel); on_flags(j, flags) */
    template <int N, typename F>
    SMH_MEMBER void next_fn(uint32_t (&row)[N], const uint32_t (&x)[N], int k, const void *tab, F &&on_flags) const
    {
        uint32_t c[N], t[N], deepest = 0;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            c[j] = smh_bfe(x[j], k == 0 ? 7 : 23, 5);
            t[j] = read_full(tab, full_addr(row[j], c[j]));
            deepest = deepest > row[j] ? deepest : row[j];
        }
        if (SMH_UNLIKELY(SMH_WAVE_ANY(deepest >= nf))) {
#pragma unroll
            for (int j = 0; j < N; ++j) t[j] = resolve(row[j], c[j], t[j], tab, [&](uint32_t f) { on_flags(j, f); });
        }
#pragma unroll
        for (int j = 0; j < N; ++j) row[j] = t[j];
    }
    /* generic form for the halo steps: row | flags << 16 */
    SMH_MEMBER uint32_t next(uint32_t e, uint32_t x, int k, const void *tab) const
    {
        uint32_t f = 0;
        const uint32_t r = next_f(e & 0xFFFFu, x, k, tab, [&](uint32_t a) { f = a; });
        return r | (f << 16);
    }
    SMH_MEMBER uint32_t flags(uint32_t e) const { return e >> 16; }
    SMH_MEMBER uint32_t row(uint32_t e) const { return e & 0xFFFFu; }
    SMH_MEMBER uint32_t any(uint32_t e) const { return e >> 16; }
};
typedef smh_fmt_s2h_t<true> smh_fmt_s2h;       /* safe everywhere */
typedef smh_fmt_s2h_t<false> smh_fmt_s2h_oob;  /* needs smh_lds_oob_reads_zero() on the device */

template <typename FMT, int HC, int NCH, bool EXACT, int SW = 16> struct smh_ac_scan_ctx {
    FMT fmt;
    const void *tab; /* LDS: depth-K automaton */
    int halo;        /* K - 1 */
    const smh_ac_df *df;
    const uint8_t *text;
    const uint64_t *a;    /* segment offsets of the NCH chains */
    const uint32_t *tail; /* 4*HC words, same in every lane: the bytes that follow the wave-chunk */
    const smh_ac_verify_ctx *V;
    smh_ac_queue *Q;
    uint32_t *hmask; /* bit-recording modes (stride-2 candidates, positions): halo flags are OR-ed in here
                      * (NCH words, bit = halo byte index), else NULL */
};

/* queue the candidates flagged by entry `e` (reached from `prev`) for the byte (pair) at `pos` */
template <typename FMT, int HC, int NCH, bool EXACT, int SW>
SMH_LANE void smh_ac_emit_flags(const smh_ac_scan_ctx<FMT, HC, NCH, EXACT, SW> &c, uint32_t f, uint32_t prev, uint32_t e,
                                uint64_t pos)
{
    if (FMT::STRIDE == 1) {
        smh_ac_emit<false>(*c.Q, *c.V, f != 0, pos, c.fmt.row(e), SMH_CAND_ROW);
    } else {
        /* first symbol of the pair: the depth-K row is not in the entry -> resolved lazily */
        smh_ac_emit<false>(*c.Q, *c.V, (f & 1u) != 0, pos, c.fmt.row(prev), SMH_CAND_LAZY);
        smh_ac_emit<false>(*c.Q, *c.V, (f & 2u) != 0, pos + 1u, c.fmt.row(e), SMH_CAND_ROW);
    }
}

/* one step of all chains with full flag handling: the halo steps and the replay of a flagged piece */
template <typename FMT, int HC, int NCH, bool EXACT, int SW>
SMH_LANE void smh_ac_step_full(const smh_ac_scan_ctx<FMT, HC, NCH, EXACT, SW> &c, const uint32_t (&x)[NCH], int k,
                               const uint64_t (&pos)[NCH], bool second_valid, uint32_t (&e)[NCH], uint32_t &cnt,
                               int hbit = 0)
{
    if constexpr (FMT::SPARSE) {
        /* hybrid image: the state stays the plain row id and flags arrive through the callback, from the rare
         * compact-row resolution only -- the common halo step carries no flag arithmetic either (packing row | flags
         * << 16 and taking it apart again was 5 VALU per chain and halo step) */
        if (c.hmask || EXACT) {
            if constexpr (NCH > 1) {
                c.fmt.next_fn(e, x, k, c.tab, [&](int j, uint32_t f) {
                    if (!second_valid) f &= 1u;
                    if (c.hmask) c.hmask[j] |= f << hbit;
                    else cnt += (uint32_t)__builtin_popcount(f);
                });
                return;
            }
#pragma unroll
            for (int j = 0; j < NCH; ++j)
                e[j] = c.fmt.next_f(c.fmt.row(e[j]), x[j], k, c.tab, [&](uint32_t f) {
                    if (!second_valid) f &= 1u;
                    if (c.hmask) c.hmask[j] |= f << hbit;
                    else cnt += (uint32_t)__builtin_popcount(f);
                });
            return;
        }
    }
    uint32_t f[NCH], prev[NCH];
    uint32_t anyf = 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        prev[j] = e[j];
        e[j] = c.fmt.next(e[j], x[j], k, c.tab);
        f[j] = c.fmt.flags(e[j]);
        if (FMT::STRIDE == 2 && !second_valid) f[j] &= 1u;
        anyf |= f[j];
    }
    if (c.hmask) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) c.hmask[j] |= f[j] << hbit;
    } else if (EXACT) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) cnt += (uint32_t)__builtin_popcount(f[j]);
    } else if (SMH_WAVE_ANY(anyf != 0)) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) smh_ac_emit_flags(c, f[j], prev[j], e[j], pos[j]);
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
template <int H, typename FMT, int HC, int NCH, bool EXACT, int SW>
SMH_LANE bool smh_ac_halo_step(const smh_ac_scan_ctx<FMT, HC, NCH, EXACT, SW> &c, const uint32_t (&w)[NCH][SW],
                               uint32_t (&hx)[NCH], uint32_t (&e)[NCH], uint32_t &cnt)
{
    if (H % FMT::STRIDE != 0) return true; /* stride 2 consumes bytes H and H+1 at even H */
    if (H >= c.halo) return false;
    const uint32_t need = c.df->v[H + 1];
    uint32_t deepest = 0; /* one compare for the lane's chains: v_max3 on three */
#pragma unroll
    for (int j = 0; j < NCH; ++j) deepest = deepest > c.fmt.row(e[j]) ? deepest : c.fmt.row(e[j]);
    if (!SMH_WAVE_ANY(deepest >= need)) return false;
    if ((H & 3) == 0) {
        /* next halo dword: the neighbour lane's segment word H/4 (all lanes active here: every
         * branch above is wave-uniform); lane 63 takes the next chain's lane 0, or the tail */
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const uint32_t edge = j + 1 < NCH ? smh_first_lane(w[j + 1 < NCH ? j + 1 : j][H >> 2]) : c.tail[H >> 2];
            hx[j] = c.fmt.prep(smh_next_lane_word(w[j][H >> 2], edge, c.text, c.a[j] + 4u * SW + (uint64_t)H));
        }
    }
    uint64_t pos[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) pos[j] = c.a[j] + 4u * SW + (uint64_t)H;
    smh_ac_step_full(c, hx, (H & 3) / FMT::STRIDE, pos, H + 1 < c.halo, e, cnt, H);
    return true;
}

template <typename FMT, int HC, int NCH, bool EXACT, int SW, int... Hs>
SMH_LANE void smh_ac_halo_all(const smh_ac_scan_ctx<FMT, HC, NCH, EXACT, SW> &c, const uint32_t (&w)[NCH][SW],
                              uint32_t (&e)[NCH], uint32_t &cnt, std::integer_sequence<int, Hs...>)
{
    uint32_t hx[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) hx[j] = 0;
    (void)(smh_ac_halo_step<Hs>(c, w, hx, e, cnt) && ...);
}

/* the 16*HC bytes that follow a wave-chunk (the post-halo of its last lane), loaded by every lane
 * from the same address together with the segments, so that the halo steps touch no memory */
template <int HC>
SMH_LANE void smh_ac_load_tail(const uint8_t *p, uint32_t (&t)[4 * HC])
{
#pragma unroll
    for (int q = 0; q < HC; ++q) {
        const smh_u32x4 v = smh_load16(p + 16u * q);
        t[4 * q + 0] = v.v[0];
        t[4 * q + 1] = v.v[1];
        t[4 * q + 2] = v.v[2];
        t[4 * q + 3] = v.v[3];
    }
}

template <int NCH, int SW>
SMH_LANE void smh_ac_load_segments(const uint8_t *text, const uint64_t (&a)[NCH], uint32_t (&w)[NCH][SW])
{
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int q = 0; q < SW / 4; ++q) {
            const smh_u32x4 t = smh_load16(text + a[j] + 16u * q);
            w[j][4 * q + 0] = t.v[0];
            w[j][4 * q + 1] = t.v[1];
            w[j][4 * q + 2] = t.v[2];
            w[j][4 * q + 3] = t.v[3];
        }
}

/* What a segment's recorded bits become once its halo is done (shared by the byte-text and the packed-text fast paths):
 * REC  positions mode with K == m: bit = match END column, appended to the output (returns the matches appended);
 * BITS depth-cut stride-2 plans: bit = END of a K-symbol prefix, queued for the verify stage by position only. */
template <int NCH, bool REC, bool BITS>
SMH_LANE uint32_t smh_ac_finish_masks(const uint64_t (&a)[NCH], const uint32_t (&mlo)[NCH], const uint32_t (&mhi)[NCH],
                                      const uint32_t (&mhalo)[NCH], const smh_ac_verify_ctx &V, smh_ac_queue &Q)
{
    uint32_t cnt = 0;
    if (REC) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            cnt += smh_append_bits2(((uint64_t)mhi[j] << 32) | mlo[j], a[j], mhalo[j], a[j] + SMH_SEG, V.pos);
        }
    }
    if (BITS) {
        /* compaction: one queue entry per set bit, as many rounds as the busiest lane has bits.  ONE vote decides the
         * common case (no candidate anywhere in the wave's chunk): the loops below each start with a vote of
         * their own, and their code sat in the way of every chunk */
        uint32_t any_bits = 0;
#pragma unroll
        for (int j = 0; j < NCH; ++j) any_bits |= mlo[j] | mhi[j] | mhalo[j];
        if (SMH_UNLIKELY(SMH_WAVE_ANY(any_bits != 0))) {
            /* ONE emit site (a rolled loop over the 3 * NCH masks) */
            uint32_t mm[3 * NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                mm[3 * j] = mlo[j];
                mm[3 * j + 1] = mhi[j];
                mm[3 * j + 2] = mhalo[j];
            }
#pragma unroll 1
            for (int g = 0; g < 3 * NCH; ++g) {
                uint32_t msk = mm[g];
                const uint64_t base = a[g / 3] + 32u * (uint32_t)(g % 3);
                while (SMH_WAVE_ANY(msk != 0)) {
                    const bool have = msk != 0;
                    const uint32_t b = have ? (uint32_t)__builtin_ctz(msk) : 0u;
                    smh_ac_emit(Q, V, have, base + b, 0u, SMH_CAND_ROOT);
                    msk &= msk - 1u;
                }
            }
        }
    }
    return cnt;
}

/*
 * Fast path: NCH segments per lane (already in registers), each fully inside the text together
 * with 16*HC bytes after it (the caller guarantees a[j] + 64 + 16*HC <= n and 16*HC >= K-1).
 * The halo bytes come out of the neighbouring lane's registers (smh_next_lane_word).  The NCH
 * automata are independent dependency chains stepped in lock-step, so the LDS latency of one
 * hides behind the others.
 *
 * K == m: every lookup adds its match flags to the count.  K < m: candidates are rare, so the
 * inner loop only ORs the entries together; the OR is examined once per 16-byte piece with a
 * wave-wide vote, and a piece that holds a candidate anywhere in the wave is walked again from a
 * snapshot of the states, this time queueing the candidates.
 */
template <typename FMT, int HC, int NCH, bool EXACT, int SW, bool POS = false>
SMH_LANE uint32_t smh_ac_lane_fast(const FMT &fmt, const uint8_t *text, const uint64_t (&a)[NCH],
                                   const uint32_t (&w)[NCH][SW], const uint32_t (&tail)[4 * HC], const void *tab,
                                   int K, const smh_ac_df &df, const smh_ac_verify_ctx &V, smh_ac_queue &Q)
{
    uint32_t e[NCH], snap[NCH], cnt = 0, anyf = 0;
#pragma unroll
    for (int j = 0; j < NCH; ++j) e[j] = snap[j] = 0;
    /* stride 2 with K < m: candidates are too frequent for "vote and replay" (a 16-byte piece of a
     * wave holds 2048 positions); every lookup instead drops its two flag bits into a per-lane bit
     * mask (one bit per text byte) and the set bits are queued after the segment, position only --
     * the verify stage then walks the pattern from the root (SMH_CAND_ROOT). */
    constexpr bool BITS = !EXACT && FMT::STRIDE == 2 && (HC == 1 || (FMT::SPARSE && HC == 2)) && SW == 16;
    static_assert(EXACT || FMT::STRIDE == 1 || SW == 16, "stride-2 candidate recording assumes 64-byte segments");
    /* positions mode with K == m: every flag is a match; they are recorded the same way (one bit per
     * END column, halo bits separately) and appended to the output after the segment */
    constexpr bool REC = POS && EXACT;
    static_assert(!REC || (HC <= 2 && SW == 16), "match recording covers a 32-byte halo");
    uint32_t mlo[NCH], mhi[NCH], mhalo[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) mlo[j] = mhi[j] = mhalo[j] = 0;
    smh_ac_scan_ctx<FMT, HC, NCH, EXACT, SW> ctx{fmt, tab, K - 1, &df, text, a, tail, &V, &Q, (BITS || REC) ? mhalo : nullptr};
    constexpr int SPD = 4 / FMT::STRIDE; /* steps per text dword */

#pragma unroll
    for (int piece = 0; piece < SW / 4; ++piece) {
#pragma unroll
        for (int q = 4 * piece; q < 4 * piece + 4; ++q) {
            uint32_t x[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) x[j] = fmt.prep(w[j][q]);
#pragma unroll
            for (int k = 0; k < SPD; ++k) {
                if constexpr (FMT::SPARSE && NCH > 1) {
                    /* all chains of the lane in one step: one vote in the common path */
                    static_assert(EXACT || BITS, "the hybrid image records candidates as bits");
                    const int bit = 4 * q + 2 * k;
                    fmt.next_fn(e, x, k, tab, [&](int j, uint32_t f) {
                        if (EXACT && !REC)
                            cnt += (uint32_t)__builtin_popcount(f);
                        else if (bit < 32)
                            mlo[j] |= f << bit;
                        else
                            mhi[j] |= f << (bit - 32);
                    });
                    continue;
                }
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    if constexpr (FMT::SPARSE) {
                        /* flags arrive through the callback, from the rare compact-row resolution only */
                        static_assert(EXACT || BITS, "the hybrid image records candidates as bits");
                        const int bit = 4 * q + 2 * k;
                        e[j] = fmt.next_f(e[j], x[j], k, tab, [&](uint32_t f) {
                            if (EXACT && !REC)
                                cnt += (uint32_t)__builtin_popcount(f);
                            else if (bit < 32)
                                mlo[j] |= f << bit;
                            else
                                mhi[j] |= f << (bit - 32);
                        });
                        continue;
                    }
                    e[j] = fmt.next(e[j], x[j], k, tab);
                    if (EXACT && !REC) {
                        cnt = smh_popc_add(fmt.flags(e[j]), cnt);
                    } else if (BITS || REC) {
                        const int bit = 4 * q + k * FMT::STRIDE;
                        if (bit < 32)
                            mlo[j] |= fmt.flags(e[j]) << bit;
                        else
                            mhi[j] |= fmt.flags(e[j]) << (bit - 32);
                    } else {
                        anyf |= e[j];
                    }
                }
            }
        }
        if (!EXACT && !BITS) {
            if (SMH_WAVE_ANY(fmt.any(anyf) != 0)) {
                /* replay the piece from the snapshot, with flag handling */
                uint32_t r[NCH];
#pragma unroll
                for (int j = 0; j < NCH; ++j) r[j] = snap[j];
#pragma unroll
                for (int q = 4 * piece; q < 4 * piece + 4; ++q) {
                    uint32_t x[NCH];
#pragma unroll
                    for (int j = 0; j < NCH; ++j) x[j] = fmt.prep(w[j][q]);
#pragma unroll
                    for (int k = 0; k < SPD; ++k) {
                        uint64_t pos[NCH];
#pragma unroll
                        for (int j = 0; j < NCH; ++j) pos[j] = a[j] + (uint64_t)(4 * q + k * FMT::STRIDE);
                        smh_ac_step_full(ctx, x, k, pos, true, r, cnt);
                    }
                }
            }
            anyf = 0;
#pragma unroll
            for (int j = 0; j < NCH; ++j) snap[j] = e[j];
        }
    }
    smh_ac_halo_all(ctx, w, e, cnt, std::make_integer_sequence<int, 16 * HC>{});
    return cnt + smh_ac_finish_masks<NCH, REC, BITS>(a, mlo, mhi, mhalo, V, Q);
}

/* Slow path: any segment, byte loads with bounds checks, stride-1 depth-K table from HBM,
 * candidates verified on the spot.  Used for the last wave-chunk(s) of a text and for texts
 * shorter than one wave-chunk. */
SMH_LANE uint32_t smh_ac_lane_slow(const smh_ac_verify_ctx &V, uint64_t n_starts, uint64_t a, uint32_t seg_bytes = SMH_SEG)
{
    if (a >= n_starts) return 0;
    const smh_ac_cold_ctx &C = *V.cold;
    uint64_t own_end = a + seg_bytes;
    if (own_end > n_starts) own_end = n_starts;
    /* K-symbol prefixes that START in [a, own_end) END before own_end + K - 1 */
    uint64_t stop = own_end + (uint64_t)(V.K - 1);
    if (stop > V.n) stop = V.n;
    const uint32_t tmask = C.trunc1_entry_bytes == 2 ? 0x7FFFu : 0x7FFFFFFFu;
    const uint32_t tshift = C.trunc1_entry_bytes == 2 ? 15u : 31u;
    uint32_t row = 0, cnt = 0;
    for (uint64_t i = a; i < stop; ++i) {
        uint32_t c = V.text[i];
        if (c >= (uint32_t)V.sigma) c = 0;
        const uint32_t e = smh_entry_at(C.trunc1, C.trunc1_entry_bytes, (uint64_t)row * (uint32_t)V.sigma + c);
        row = e & tmask;
        if (e >> tshift) cnt += V.K >= V.m ? 1u : smh_ac_deep_walk(V, i, row, SMH_CAND_ROW);
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
SMH_LANE uint64_t smh_ac_segment_match_mask(const smh_ac_verify_ctx &V, uint64_t n_starts, uint64_t a);

/* One 4 KiB piece (64 lanes x one segment) of a wave-chunk that did not qualify for the multi-segment fast path -- the
 * text's last chunk, in practice.  A piece that still lies inside the text with its halo takes the one-segment fast path;
 * only the piece that holds the text's end walks byte by byte with bounds checks (~30 us for its 64 bytes per lane: every
 * step is a dependent load from HBM/L2).  Without this split a lane of an N-segment kernel walked all N segments of the
 * last chunk that way, one after the other: 90 us with three segments per lane, 180 us with six -- the whole launch. */
template <typename FMT, int HC, bool EXACT, int SW, bool POS>
SMH_LANE uint32_t smh_ac_piece(const FMT &fmt, const smh_ac_verify_ctx &V, uint64_t piece_base, uint32_t lane, uint64_t n_starts,
                               const void *tab, const smh_ac_df &df, smh_ac_queue &Q)
{
    constexpr uint32_t SEGB = 4u * SW;
    const uint64_t as = piece_base + (uint64_t)lane * SEGB;
    if (piece_base + 64u * SEGB + 16u * HC <= V.n) { /* wave-uniform */
        uint64_t a[1] = {as};
        uint32_t w[1][SW], tail[4 * HC];
        smh_ac_load_segments<1, SW>(V.text, a, w);
        smh_ac_load_tail<HC>(V.text + piece_base + 64u * SEGB, tail);
        return smh_ac_lane_fast<FMT, HC, 1, EXACT, SW, POS>(fmt, V.text, a, w, tail, tab, V.K, df, V, Q);
    }
    if (POS) return smh_append_bits(smh_ac_segment_match_mask(V, n_starts, as), as + (uint64_t)(V.m - 1), V.pos);
    return smh_ac_lane_slow(V, n_starts, as, SEGB);
}

/* POS: positions mode -- instead of counting, every match appends its END column to V.pos (the return
 * value is then the number of matches this lane appended; the kernels ignore it). */
template <typename FMT, int HC, int NCH, bool EXACT, int PREFETCH = SMH_PREFETCH, int SW = 16, bool POS = false>
SMH_LANE uint32_t smh_ac_thread(const FMT &fmt, uint64_t gthread, const smh_chunk_sched &S, const void *tab,
                                const smh_ac_verify_ctx &V, const smh_ac_df &df, uint64_t *queue_base, uint32_t *events_out = nullptr)
{
    if (V.n < (uint64_t)V.m) return 0;
    const uint64_t n_starts = V.n - (uint64_t)V.m + 1;
    constexpr uint32_t SEGB = 4u * SW; /* bytes per lane segment */
    const uint64_t chunk_bytes = (uint64_t)SEGB * 64u * NCH;
    const uint64_t n_chunks = (n_starts + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    const uint64_t wave = gthread >> 6;
    smh_ac_queue Q;
    Q.slots = queue_base ? queue_base + smh_uniform64(wave) * SMH_AC_QCAP : nullptr;
    Q.count = 0;
    Q.matches = 0;
    Q.events = 0;
    uint32_t cnt = 0;
    /* software pipeline: the segments of the wave's NEXT chunk are requested before the current
     * chunk is scanned, so the HBM latency of a chunk hides behind a whole chunk of lookups */
    uint32_t cur[NCH][SW], nxt[NCH][SW], cur_tail[4 * HC], nxt_tail[4 * HC];
    uint64_t k = S.take(n_chunks);
    bool cur_fast = false;
    if (k < n_chunks) {
        const uint64_t base = smh_uniform64(k * chunk_bytes); /* same for the 64 lanes of a wave */
        cur_fast = base + chunk_bytes + 16u * HC <= V.n;
        if (cur_fast) {
            uint64_t a[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) a[j] = base + ((uint64_t)j * 64u + lane) * SEGB;
            smh_ac_load_segments<NCH, SW>(V.text, a, cur);
            smh_ac_load_tail<HC>(V.text + base + chunk_bytes, cur_tail);
        }
    }
    while (k < n_chunks) {
        const uint64_t base = smh_uniform64(k * chunk_bytes);
        const uint64_t kn = S.take(n_chunks); /* taken before this chunk is scanned: its text is prefetched below */
        const uint64_t base_n = smh_uniform64(kn * chunk_bytes);
        const bool nxt_fast = kn < n_chunks && base_n + chunk_bytes + 16u * HC <= V.n;
        if (PREFETCH == 1 && nxt_fast) {
            uint64_t an[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) an[j] = base_n + ((uint64_t)j * 64u + lane) * SEGB;
            smh_ac_load_segments<NCH, SW>(V.text, an, nxt);
            smh_ac_load_tail<HC>(V.text + base_n + chunk_bytes, nxt_tail);
        }
        if (cur_fast) {
            uint64_t a[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) a[j] = base + ((uint64_t)j * 64u + lane) * SEGB;
            cnt += smh_ac_lane_fast<FMT, HC, NCH, EXACT, SW, POS>(fmt, V.text, a, cur, cur_tail, tab, V.K, df, V, Q);
        } else if (NCH > 1) {
            /* the text's last chunk: piece by piece, fast where a piece still has its halo inside the text */
            static_assert(!POS || SW == 16, "positions mode uses 64-byte segments");
#pragma unroll 1
            for (int j = 0; j < NCH; ++j)
                cnt += smh_ac_piece<FMT, HC, EXACT, SW, POS>(fmt, V, base + (uint64_t)j * 64u * SEGB, lane, n_starts, tab, df, Q);
        } else if (POS) {
            /* the text's last chunk(s): per-lane mask of matching STARTS, then the wave-level append */
            static_assert(!POS || SW == 16, "positions mode uses 64-byte segments");
            const uint64_t as = base + (uint64_t)lane * SEGB;
            cnt += smh_append_bits(smh_ac_segment_match_mask(V, n_starts, as), as + (uint64_t)(V.m - 1), V.pos);
        } else {
            cnt += smh_ac_lane_slow(V, n_starts, base + (uint64_t)lane * SEGB, SEGB);
        }
        if (nxt_fast) {
            if (PREFETCH == 1) {
#pragma unroll
                for (int j = 0; j < NCH; ++j)
#pragma unroll
                    for (int q = 0; q < SW; ++q) cur[j][q] = nxt[j][q];
#pragma unroll
                for (int q = 0; q < 4 * HC; ++q) cur_tail[q] = nxt_tail[q];
            } else {
                uint64_t an[NCH];
#pragma unroll
                for (int j = 0; j < NCH; ++j) an[j] = base_n + ((uint64_t)j * 64u + lane) * SEGB;
                smh_ac_load_segments<NCH, SW>(V.text, an, cur);
                smh_ac_load_tail<HC>(V.text + base_n + chunk_bytes, cur_tail);
            }
        }
        cur_fast = nxt_fast;
        k = kn;
    }
    if (!EXACT) smh_ac_drain(Q, V);
    if (events_out) *events_out = Q.events;
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

/* ------------------------------------------------------------------ match positions
 * The reference only ever printed positions from commented-out code (ac/ac.c:217 "match at %i"
 * with the END column); this path appends the END columns of all matches to a device buffer.
 * One lane scans the starts of one 64-byte segment with the stride-1 depth-K table read from
 * HBM/L2 (candidates verified on the spot), collecting a 64-bit mask of matching starts; the wave
 * then compacts: prefix sum of the per-lane counts, one atomic on the cursor, coalesced-ish
 * stores.  Order of the output is unspecified (sort to compare).  Entries beyond `capacity` are
 * dropped; the cursor still counts them, so cursor > capacity tells the caller to retry bigger.
 */
SMH_LANE uint64_t smh_ac_segment_match_mask(const smh_ac_verify_ctx &V, uint64_t n_starts, uint64_t a)
{
    if (a >= n_starts) return 0;
    const smh_ac_cold_ctx &C = *V.cold;
    uint64_t own_end = a + SMH_SEG;
    if (own_end > n_starts) own_end = n_starts;
    uint64_t stop = own_end + (uint64_t)(V.K - 1);
    if (stop > V.n) stop = V.n;
    const uint32_t tmask = C.trunc1_entry_bytes == 2 ? 0x7FFFu : 0x7FFFFFFFu;
    const uint32_t tshift = C.trunc1_entry_bytes == 2 ? 15u : 31u;
    uint32_t row = 0;
    uint64_t mask = 0;
    for (uint64_t i = a; i < stop; ++i) {
        uint32_t c = V.text[i];
        if (c >= (uint32_t)V.sigma) c = 0;
        const uint32_t e = smh_entry_at(C.trunc1, C.trunc1_entry_bytes, (uint64_t)row * (uint32_t)V.sigma + c);
        row = e & tmask;
        if (e >> tshift) {
            const uint32_t hit = V.K >= V.m ? 1u : smh_ac_deep_walk(V, i, row, SMH_CAND_ROW);
            /* the K-symbol prefix that ends at i starts at i - K + 1, inside [a, own_end) */
            if (hit) mask |= 1ull << (i + 1 - (uint64_t)V.K - a);
        }
    }
    return mask;
}

/* append END columns (start + m - 1) of the set bits; returns the number of matches */
SMH_LANE uint32_t smh_append_positions(uint64_t mask, uint64_t a, int m, uint64_t *positions, uint64_t capacity,
                                       uint64_t *cursor)
{
    const uint32_t mine = (uint32_t)__builtin_popcountll(mask);
    uint64_t slot = smh_wave_reserve(cursor, mine);
    while (mask) {
        const int b = __builtin_ctzll(mask);
        mask &= mask - 1;
        if (slot < capacity) positions[slot] = a + (uint64_t)b + (uint64_t)(m - 1);
        ++slot;
    }
    return mine;
}

SMH_LANE void smh_ac_positions_thread(uint64_t gthread, uint64_t nthreads, const smh_ac_verify_ctx &V,
                                      uint64_t *positions, uint64_t capacity, uint64_t *cursor)
{
    if (V.n < (uint64_t)V.m) return;
    const uint64_t n_starts = V.n - (uint64_t)V.m + 1;
    const uint64_t n_segs = (n_starts + SMH_SEG - 1) / SMH_SEG;
    const uint64_t n_rounds = (n_segs + nthreads - 1) / nthreads; /* same trip count for every lane of a wave */
    for (uint64_t r = 0; r < n_rounds; ++r) {
        const uint64_t a = (r * nthreads + gthread) * SMH_SEG;
        const uint64_t mask = smh_ac_segment_match_mask(V, n_starts, a);
        smh_append_positions(mask, a, V.m, positions, capacity, cursor);
    }
}

#endif
