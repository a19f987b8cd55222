/*
 * csrc/key_lane.h -- lane code of the key engine (key_hash.h): what one lane does with its 64-byte text segment.
 *
 * A lane owns the 64 END columns of its segment.  It keeps the code of the last symbols in one or two registers
 * (code = code << bits | symbol: one v_lshl_or, for 64-bit keys a v_alignbit in front of it), primed with the 16 * HP
 * bytes in front of the segment -- the previous lane's last registers (DPP wave_shr:1), lane 0 the wave-uniform bytes
 * in front of the wave-chunk -- and asks for every column: is the key (the code's low m * bits bits) in slot h1(key) of
 * table 1 or slot h2(key) of table 2?  Two independent LDS reads per column, nothing that depends on what the text is or
 * on what was found: ac/ac.c:207-219's loop with the state replaced by the window itself.
 *
 * Compiled for the GPU (key_kernels.hip) and, with SMH_HOST_EMU, for the CPU lane emulator (tests/emu).
 */
#ifndef SMH_KEY_LANE_H
#define SMH_KEY_LANE_H

#include "lane_common.h"
#include "wm_lane.h" /* smh_prev_lane_word, smh_lds_u32x2 */
#include "key_hash.h"

struct smh_key_code { uint32_t lo, hi, h; }; /* h: 64-bit keys only, the window's rolled hash (key_hash.h smh_key_poly_*) */

/* acc += the number of lanes of the wave whose `hit` is set, on the scalar unit (one s_bcnt1 + s_add per column instead of a
 * v_cndmask / v_addc pair: the loop is VALU-issue bound); acc is wave-uniform.  The emulation runs one lane at a time. */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
/* (one ballot per compare, OR-ed on the scalar unit: the ballot of an OR of two compares is lowered through v_cndmask + v_cmp_ne) */
SMH_LANE void smh_key_count(uint32_t &acc, bool hit1, bool hit2)
{
    acc += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit1) | __builtin_amdgcn_ballot_w64(hit2));
}
SMH_LANE uint32_t smh_key_count_mine(uint32_t acc) { return (threadIdx.x & 63u) == 0u ? acc : 0u; }
#else
SMH_LANE void smh_key_count(uint32_t &acc, bool hit1, bool hit2) { acc += (hit1 || hit2) ? 1u : 0u; }
SMH_LANE uint32_t smh_key_count_mine(uint32_t acc) { return acc; }
#endif

/* KC = the key class (smh_key_params.wide): 0 = 32-bit keys, 1 = 64-bit keys in 8-byte slots, 2 = quotient keys (33..42 bits,
 * the low 32 in a 4-byte slot, the high bits added to the slot number) */
template <int KC, bool FULL = false>
SMH_LANE void smh_key_roll(smh_key_code &c, uint32_t sym, uint32_t bits, const smh_key_params &K)
{
    if constexpr (KC == 1) {
        /* the symbol that leaves the window sits in the code's top symbol (m * bits > 42 and bits <= 8: always in the high register) */
        const uint32_t out = FULL ? c.hi >> (32u - bits) : smh_bfe(c.hi, (uint32_t)K.m * bits - bits - 32u, bits);
        c.h = smh_key_poly_out(smh_key_poly_in(c.h, sym, K.fold[0]), out, K.fold[1]);
    }
    if constexpr (KC != 0) {
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
        c.hi = __builtin_amdgcn_alignbit(c.hi, c.lo, 32u - bits);
#else
        c.hi = (c.hi << bits) | (c.lo >> (32u - bits));
#endif
    }
    c.lo = (c.lo << bits) | sym;
}

/* the key of code c and the byte offsets of its two slots in the image */
struct smh_key_probe { uint32_t klo, khi, o1, o2; };
/* FULL: the key fills its slot (m * bits == 32 or 64) -- the rolling code IS the key, no mask */
template <int KC, bool FULL = false>
SMH_LANE smh_key_probe smh_key_address(const smh_key_code &c, const smh_key_params &K)
{
    smh_key_probe p;
    p.klo = FULL || KC == 2 ? c.lo : (c.lo & K.mask_lo);
    p.khi = KC != 0 ? (FULL ? c.hi : (c.hi & K.mask_hi)) : 0u;
    const uint32_t f = KC == 1 ? c.h : p.klo;
    const uint32_t h1 = KC == 1 ? f : smh_key_mul24(f, K.mul[0]) + smh_key_mul24(f >> 8, K.mul[1]), h2 = smh_key_mul24(h1, K.mul[2]);
    const uint32_t ns = K.slots << 8;
    if constexpr (KC == 2) { /* the key's high bits move the slot (one v_add_lshl per table instead of the shift) */
        p.o1 = (smh_key_mulhi24(h1, ns) + p.khi) << 2;
        p.o2 = ((smh_key_mulhi24(h2, ns) + p.khi) << 2) + K.base2;
    } else {
        p.o1 = smh_key_mulhi24(h1, ns) << (KC == 1 ? 3 : 2);
        p.o2 = (smh_key_mulhi24(h2, ns) << (KC == 1 ? 3 : 2)) + K.base2;
    }
    return p;
}
/* the two slots' contents */
struct smh_key_slots2 { uint32_t a0, a1, b0, b1; };
template <int KC>
SMH_LANE smh_key_slots2 smh_key_read(const smh_key_probe &p, const void *tab)
{
    smh_key_slots2 r = {0u, 0u, 0u, 0u};
    if constexpr (KC == 1) {
        smh_lds_u32x2(tab, p.o1, r.a0, r.a1);
        smh_lds_u32x2(tab, p.o2, r.b0, r.b1);
    } else {
        r.a0 = smh_lds_u32(tab, p.o1);
        r.b0 = smh_lds_u32(tab, p.o2);
    }
    return r;
}
/* slot t of the probe holds the key */
template <int KC, int T>
SMH_LANE bool smh_key_slot_is(const smh_key_probe &p, const smh_key_slots2 &r)
{
    if constexpr (KC == 1) return ((((uint64_t)(T ? r.b1 : r.a1)) << 32) | (T ? r.b0 : r.a0)) == ((((uint64_t)p.khi) << 32) | p.klo);
    return (T ? r.b0 : r.a0) == p.klo;
}
template <int KC>
SMH_LANE bool smh_key_decide(const smh_key_probe &p, const smh_key_slots2 &r)
{
    /* no short-circuit: as `||` the compiler puts the second slot's READ behind a branch on the first compare */
    if constexpr (KC == 1) {
        const uint64_t key = ((uint64_t)p.khi << 32) | p.klo, a = ((uint64_t)r.a1 << 32) | r.a0, b = ((uint64_t)r.b1 << 32) | r.b0;
        return (a == key) | (b == key);
    }
    return (r.a0 == p.klo) | (r.b0 == p.klo);
}
/* 1 when the window whose code is c is a pattern.  `tab` = the image (LDS offset 0 on the GPU). */
template <int KC>
SMH_LANE uint32_t smh_key_test(const smh_key_code &c, const void *tab, const smh_key_params &K)
{
    const smh_key_probe p = smh_key_address<KC>(c, K);
    return smh_key_decide<KC>(p, smh_key_read<KC>(p, tab)) ? 1u : 0u;
}

/* fast path: the 64 END columns of the segment at a (a >= 16 * HP >= m - 1, a + 64 <= n); edge = the 16 * HP bytes in
 * front of the wave-chunk (wave-uniform), what lane 0 primes with.  The columns go four at a time (one text dword): four
 * keys and their eight slot addresses, eight LDS reads in flight together, then the eight compares. */
template <int KC, int HP, bool POS, bool FULL = false>
SMH_LANE uint32_t smh_key_lane_fast(const uint8_t *text, uint64_t a, const uint32_t (&w)[16], const uint32_t (&edge)[4 * HP],
                                    const void *tab, const smh_key_params &K, const smh_pos_out *po)
{
    const uint32_t bits = (uint32_t)K.bits;
    smh_key_code c = {0u, 0u, 0u};
#pragma unroll
    for (int q = 0; q < 4 * HP; ++q) {
        const uint32_t pw = smh_prev_lane_word(w[16 - 4 * HP + q], edge[q], text, a - 16u * HP + 4u * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) smh_key_roll<KC, FULL>(c, smh_bfe(pw, 8u * k, bits), bits, K);
    }
    uint32_t cnt = 0, mlo = 0, mhi = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        smh_key_probe p[4];
        smh_key_slots2 r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            smh_key_roll<KC, FULL>(c, smh_bfe(w[q], 8u * k, bits), bits, K);
            p[k] = smh_key_address<KC, FULL>(c, K);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = smh_key_read<KC>(p[k], tab);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (POS) {
                const bool hit = smh_key_decide<KC>(p[k], r[k]);
                if (q < 8) mlo |= (hit ? 1u : 0u) << (4 * q + k);
                else mhi |= (hit ? 1u : 0u) << (4 * (q - 8) + k);
            } else {
                smh_key_count(cnt, smh_key_slot_is<KC, 0>(p[k], r[k]), smh_key_slot_is<KC, 1>(p[k], r[k]));
            }
        }
    }
    if constexpr (POS) return smh_append_bits(((uint64_t)mhi << 32) | mlo, a, *po);
    return smh_key_count_mine(cnt);
}

/* bounds-checked path: the END columns [max(a, m - 1), min(a + 64, n)) byte by byte from memory */
template <int KC>
SMH_LANE uint32_t smh_key_lane_slow(const uint8_t *text, uint64_t n, uint64_t a, const void *tab, const smh_key_params &K,
                                    uint64_t *match_mask = nullptr)
{
    if (match_mask) *match_mask = 0;
    if (a >= n) return 0;
    uint64_t end = a + SMH_SEG;
    if (end > n) end = n;
    uint64_t e0 = a;
    if (e0 < (uint64_t)(K.m - 1)) e0 = (uint64_t)(K.m - 1);
    if (e0 >= end) return 0;
    const uint32_t bits = (uint32_t)K.bits, smask = (1u << bits) - 1u;
    smh_key_code c = {0u, 0u, 0u};
    uint32_t cnt = 0;
    for (uint64_t i = e0 - (uint64_t)(K.m - 1); i < e0; ++i) smh_key_roll<KC>(c, text[i] & smask, bits, K);
    for (uint64_t e = e0; e < end; ++e) {
        smh_key_roll<KC>(c, text[e] & smask, bits, K);
        const uint32_t hit = smh_key_test<KC>(c, tab, K);
        cnt += hit;
        if (match_mask && hit) *match_mask |= 1ull << (e - a);
    }
    return cnt;
}

template <int KC, int HP, bool POS, bool FULL = false>
SMH_LANE uint32_t smh_key_thread(uint64_t gthread, const smh_chunk_sched &S, const uint8_t *text, uint64_t n, const void *tab,
                                 const smh_key_params &K, const smh_pos_out *po = nullptr)
{
    if (n < (uint64_t)K.m) return 0;
    const uint64_t chunk_bytes = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (n + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    uint32_t cnt = 0;
    uint32_t cur[16], edge[4 * HP];
    /* chunk 0 has no text in front of it and columns without a whole window; the last chunk may end inside a segment */
    auto is_fast = [&](uint64_t kk) { return kk >= 1 && kk < n_chunks && (kk + 1) * chunk_bytes <= n; };
    auto load = [&](uint64_t kk) {
        const uint64_t base = smh_uniform64(kk * chunk_bytes);
        const uint8_t *p = text + base + (uint64_t)lane * SMH_SEG;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const smh_u32x4 t = smh_load16(p + 16u * q);
            cur[4 * q + 0] = t.v[0];
            cur[4 * q + 1] = t.v[1];
            cur[4 * q + 2] = t.v[2];
            cur[4 * q + 3] = t.v[3];
        }
#pragma unroll
        for (int q = 0; q < HP; ++q) { /* the bytes in front of the wave-chunk, same address in every lane */
            const smh_u32x4 t = smh_load16(text + base - 16u * (uint32_t)(HP - q));
            edge[4 * q + 0] = t.v[0];
            edge[4 * q + 1] = t.v[1];
            edge[4 * q + 2] = t.v[2];
            edge[4 * q + 3] = t.v[3];
        }
    };
    uint64_t k = S.take(n_chunks);
    while (k < n_chunks) {
        const uint64_t a = smh_uniform64(k * chunk_bytes) + (uint64_t)lane * SMH_SEG;
        if (is_fast(k)) {
            load(k);
            cnt += smh_key_lane_fast<KC, HP, POS, FULL>(text, a, cur, edge, tab, K, po);
        } else if (POS) {
            uint64_t mm;
            smh_key_lane_slow<KC>(text, n, a, tab, K, &mm);
            cnt += smh_append_bits(mm, a, *po);
        } else {
            cnt += smh_key_lane_slow<KC>(text, n, a, tab, K);
        }
        k = S.take(n_chunks);
    }
    return cnt;
}

/* ------------------------------------------------------------------ the bucket image (key_hash.h "The bucket image"; round 6)
 *
 * Per column: the image H rolled by one symbol (v_mul_u32_u24 on the symbol's byte + v_lshl_add), F for windows longer than the
 * image (one v_lshl_add with the H of R columns ago, kept in registers: the loop is unrolled, the delay line is indexed by
 * constants), the bucket's byte offset (shift + and), ONE ds_read_b64, three compares (slot 0, slot 1, sentinel) whose results
 * go to the scalar unit as lane masks.  Lanes that read a sentinel -- the bucket held three keys or more -- look into the
 * overflow table behind a wave-uniform branch. */
struct smh_keyb_slots { uint32_t s0, s1; };
SMH_LANE smh_keyb_slots smh_keyb_read(const void *tab, uint32_t off)
{
    smh_keyb_slots r;
    smh_lds_u32x2(tab, off, r.s0, r.s1);
    return r;
}
/* the four slots of an overflow bucket hold H? */
SMH_LANE bool smh_keyb_overflow_has(const void *tab, uint32_t F, uint32_t H, const smh_key_params &K)
{
    const uint32_t off = smh_keyb_off2(F, &K);
    uint32_t a, b, c, d;
    smh_lds_u32x2(tab, off, a, b);
    smh_lds_u32x2(tab, off + 8u, c, d);
    return (a == H) | (b == H) | (c == H) | (d == H);
}
/* the whole test for one window, any lane by itself: the bounds-checked path and the model the fast path must equal */
SMH_LANE bool smh_keyb_hit(const void *tab, uint32_t H, uint32_t Hold, const smh_key_params &K)
{
    const uint32_t F = smh_keyb_mix(H, Hold, &K);
    const smh_keyb_slots r = smh_keyb_read(tab, smh_keyb_off1(F, &K));
    if (r.s0 != K.bk_sentinel) return (r.s0 == H) | (r.s1 == H);
    return (r.s1 == H) || smh_keyb_overflow_has(tab, F, H, K); /* (H is never the sentinel here: key_hash.h) */
}

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
typedef uint64_t smh_keyb_mask;
SMH_LANE smh_keyb_mask smh_keyb_ballot(bool x) { return __builtin_amdgcn_ballot_w64(x); }
SMH_LANE uint32_t smh_keyb_popc(smh_keyb_mask x) { return (uint32_t)__builtin_popcountll(x); }
SMH_LANE bool smh_keyb_mine(smh_keyb_mask x) { return (x >> (threadIdx.x & 63u)) & 1u; }
#else
typedef uint32_t smh_keyb_mask; /* the emulation runs one lane at a time: a mask is that lane's bit */
SMH_LANE smh_keyb_mask smh_keyb_ballot(bool x) { return x ? 1u : 0u; }
SMH_LANE uint32_t smh_keyb_popc(smh_keyb_mask x) { return x; }
SMH_LANE bool smh_keyb_mine(smh_keyb_mask x) { return x != 0; }
#endif

#ifndef SMH_KEYB_GROUP
#define SMH_KEYB_GROUP 2
#endif
/* The overflow path is DEFERRED (counting kernels on the GPU).  Looking into the overflow table the moment a lane reads a sentinel
 * costs the whole wave ten instructions and an LDS round trip for the one lane in 64 that needs it -- 1.4 % of the lanes, hence six
 * columns in ten: measured 0.57 against 0.34 ms/GiB without (profiles/r06_final/notes/ab_key_bucket_image.log).  Instead the lanes that
 * read a sentinel append their (H, F) to a queue of the wave in LDS (ballot + mbcnt: the wave's entries are contiguous), and the
 * queue is drained 64 entries per step -- every lane one entry, one overflow-table read in flight per lane -- whenever 64 more
 * might not fit, and at the end of the segment.  Slot 1 of the crowded bucket has been compared in line; a window is a key in one
 * place at most, so the drain's hits are simply added. */
#define SMH_KEYB_QCAP 96u /* entries per wave: 4 bytes each (H), 8 where F != H */
#define SMH_KEYB_QBYTES(R) (16u * SMH_KEYB_QCAP * ((R) > 0 ? 8u : 4u)) /* per 1024-thread workgroup */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
struct smh_keyb_queue { uint32_t off; uint32_t count; uint32_t hits; }; /* off: LDS byte offset of this wave's entries; count, hits: wave-uniform */
template <int R>
SMH_LANE void smh_keyb_drain(smh_keyb_queue &Q, const void *tab, const smh_key_params &K)
{
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t base = 0; base < Q.count; base += 64u) { /* wave-uniform */
        bool more = false;
        if (base + lane < Q.count) {
            const uint32_t H = smh_lds_u32(tab, Q.off + 4u * (base + lane));
            const uint32_t F = R > 0 ? smh_lds_u32(tab, Q.off + 4u * SMH_KEYB_QCAP + 4u * (base + lane)) : H;
            more = smh_keyb_overflow_has(tab, F, H, K);
        }
        Q.hits += smh_keyb_popc(smh_keyb_ballot(more));
    }
    Q.count = 0u;
}
/* the lanes of `mf` (a sentinel in slot 0) queue their window; mf != 0, wave-uniform.  false: the queue is full (the caller looks
 * into the overflow table in line -- only a text that keeps hitting crowded buckets gets there) */
template <int R>
SMH_LANE bool smh_keyb_defer(smh_keyb_queue &Q, smh_keyb_mask mf, bool mine, uint32_t H, uint32_t F)
{
    const uint32_t more = smh_keyb_popc(mf);
    if (Q.count + more > SMH_KEYB_QCAP) return false;
    const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(mf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mf, 0u));
    if (mine) {
        const uint32_t at = Q.off + 4u * (Q.count + before);
        *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(at) = H;
        if constexpr (R > 0) *reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(at + 4u * SMH_KEYB_QCAP) = F;
    }
    Q.count += more;
    return true;
}
#endif
/* R = the delay line's length (smh_key_params.bk_old: 0, 15 or 6); the 16 * HP bytes in front of the segment prime image and line */
template <int R, int HP, bool POS>
SMH_LANE uint32_t smh_keyb_lane_fast(const uint8_t *text, uint64_t a, const uint32_t (&w)[16], const uint32_t (&edge)[4 * HP],
                                     const void *tab, const smh_key_params &K, const smh_pos_out *po, uint32_t queue_off)
{
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    smh_keyb_queue Q = {queue_off, 0u, 0u};
#else
    (void)queue_off;
#endif
    static_assert(R <= 16 * HP, "the halo fills the delay line");
    const uint32_t bits = (uint32_t)K.bits, mul = K.bk_mul, smask = K.bk_symmask;
    uint32_t H = 0;
    uint32_t line[R > 0 ? R : 1];
#pragma unroll
    for (int q = 0; q < 4 * HP; ++q) {
        const uint32_t pw = smh_prev_lane_word(w[16 - 4 * HP + q], edge[q], text, a - 16u * HP + 4u * q) & smask;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            H = smh_keyb_roll(H, smh_byte_of(pw, k), bits, mul);
            if constexpr (R > 0) line[(4 * q + k) % R] = H;
        }
    }
    uint32_t cnt = 0, mlo = 0, mhi = 0;
    constexpr int G = SMH_KEYB_GROUP; /* text dwords per step: 4 * G columns' reads in flight, then the compares, then ONE look at the sentinels */
#pragma unroll
    for (int q = 0; q < 16; q += G) {
        uint32_t Hk[4 * G], Fk[4 * G];
        smh_keyb_slots r[4 * G];
#pragma unroll
        for (int j = 0; j < 4 * G; ++j) {
            const uint32_t ww = w[q + j / 4] & smask;
            H = smh_keyb_roll(H, smh_byte_of(ww, j % 4), bits, mul);
            Hk[j] = Fk[j] = H;
            if constexpr (R > 0) {
                const int at = (16 * HP + 4 * q + j) % R; /* a constant once the loops are unrolled; this slot took its value R columns ago */
                Fk[j] = (line[at] << K.bk_q) + H;
                line[at] = H;
            }
        }
#pragma unroll
        for (int j = 0; j < 4 * G; ++j) r[j] = smh_keyb_read(tab, smh_keyb_off1(Fk[j], &K));
        /* a sentinel is no probe's H (key_hash.h; key_host.c keeps the one bucket where it could be from overflowing), so slot 0's
         * compare needs no mask: a lane that reads a sentinel can only hit in slot 1 or in the overflow table */
        /* ... and a window is a key in ONE place at most: hits of the two slots and hits of the overflow table are counted
         * separately, and no lane mask outlives its column (masks are scalar register pairs: eight columns' worth spill) */
        uint32_t bits4 = 0; /* positions mode: this lane's hits of the step's columns */
        smh_keyb_mask any = 0;
#pragma unroll
        for (int j = 0; j < 4 * G; ++j) {
            if constexpr (POS) bits4 |= (((r[j].s0 == Hk[j]) | (r[j].s1 == Hk[j])) ? 1u : 0u) << j;
            else cnt += smh_keyb_popc(smh_keyb_ballot(r[j].s0 == Hk[j]) | smh_keyb_ballot(r[j].s1 == Hk[j]));
            any |= smh_keyb_ballot(r[j].s0 == K.bk_sentinel);
        }
        if (any) { /* wave-uniform: some lane of the wave read a sentinel in one of the step's columns (the builder keeps that rare) */
#pragma unroll
            for (int j = 0; j < 4 * G; ++j) {
                const bool crowded = r[j].s0 == K.bk_sentinel;
                const smh_keyb_mask mf = smh_keyb_ballot(crowded);
                if (!mf) continue; /* wave-uniform */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
                if constexpr (!POS) {
                    if (smh_keyb_defer<R>(Q, mf, crowded, Hk[j], Fk[j])) continue;
                }
#endif
                bool more = false;
                if (crowded) more = smh_keyb_overflow_has(tab, Fk[j], Hk[j], K);
                if constexpr (POS) bits4 |= (more ? 1u : 0u) << j;
                else cnt += smh_keyb_popc(smh_keyb_ballot(more));
            }
        }
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
        if constexpr (!POS) {
            if (q + G == 8) smh_keyb_drain<R>(Q, tab, K); /* mid-segment; the other drain is at its end */
        }
#endif
        if constexpr (POS) {
            if (4 * q < 32) mlo |= bits4 << (4 * q);
            else mhi |= bits4 << (4 * q - 32);
        }
    }
    if constexpr (POS) return smh_append_bits(((uint64_t)mhi << 32) | mlo, a, *po);
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
    smh_keyb_drain<R>(Q, tab, K);
    cnt += Q.hits;
#endif
    return smh_key_count_mine(cnt);
}

/* bounds-checked path: END columns [max(a, m - 1), min(a + 64, n)) byte by byte from memory; a second image lags bk_old symbols behind */
SMH_LANE uint32_t smh_keyb_lane_slow(const uint8_t *text, uint64_t n, uint64_t a, const void *tab, const smh_key_params &K,
                                     uint64_t *match_mask = nullptr)
{
    if (match_mask) *match_mask = 0;
    if (a >= n) return 0;
    uint64_t end = a + SMH_SEG;
    if (end > n) end = n;
    uint64_t e0 = a;
    if (e0 < (uint64_t)(K.m - 1)) e0 = (uint64_t)(K.m - 1);
    if (e0 >= end) return 0;
    const uint32_t bits = (uint32_t)K.bits, smask = (1u << bits) - 1u, R = K.bk_old;
    uint32_t H = 0, Hold = 0, cnt = 0;
    const uint64_t first = e0 - (uint64_t)(K.m - 1); /* the first window's first symbol */
    for (uint64_t i = first; i < e0; ++i) H = smh_keyb_roll(H, text[i] & smask, bits, K.bk_mul);
    if (R) for (uint64_t i = first; i + R < e0; ++i) Hold = smh_keyb_roll(Hold, text[i] & smask, bits, K.bk_mul);
    for (uint64_t e = e0; e < end; ++e) {
        H = smh_keyb_roll(H, text[e] & smask, bits, K.bk_mul);
        if (R) Hold = smh_keyb_roll(Hold, text[e - R] & smask, bits, K.bk_mul);
        const bool hit = smh_keyb_hit(tab, H, Hold, K);
        cnt += hit ? 1u : 0u;
        if (match_mask && hit) *match_mask |= 1ull << (e - a);
    }
    return cnt;
}

template <int R, int HP, bool POS>
SMH_LANE uint32_t smh_keyb_thread(uint64_t gthread, const smh_chunk_sched &S, const uint8_t *text, uint64_t n, const void *tab,
                                  const smh_key_params &K, const smh_pos_out *po = nullptr, uint32_t queue_off = 0u)
{
    if (n < (uint64_t)K.m) return 0;
    const uint64_t chunk_bytes = (uint64_t)SMH_SEG * 64u;
    const uint64_t n_chunks = (n + chunk_bytes - 1) / chunk_bytes;
    const uint32_t lane = (uint32_t)(gthread & 63u);
    uint32_t cnt = 0;
    uint32_t cur[16], edge[4 * HP];
    auto is_fast = [&](uint64_t kk) { return kk >= 1 && kk < n_chunks && (kk + 1) * chunk_bytes <= n; };
    uint64_t k = S.take(n_chunks);
    while (k < n_chunks) {
        const uint64_t base = smh_uniform64(k * chunk_bytes);
        const uint64_t a = base + (uint64_t)lane * SMH_SEG;
        if (is_fast(k)) {
            const uint8_t *p = text + a;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const smh_u32x4 t = smh_load16(p + 16u * q);
                cur[4 * q + 0] = t.v[0]; cur[4 * q + 1] = t.v[1]; cur[4 * q + 2] = t.v[2]; cur[4 * q + 3] = t.v[3];
            }
#pragma unroll
            for (int q = 0; q < HP; ++q) { /* the bytes in front of the wave-chunk, same address in every lane */
                const smh_u32x4 t = smh_load16(text + base - 16u * (uint32_t)(HP - q));
                edge[4 * q + 0] = t.v[0]; edge[4 * q + 1] = t.v[1]; edge[4 * q + 2] = t.v[2]; edge[4 * q + 3] = t.v[3];
            }
            cnt += smh_keyb_lane_fast<R, HP, POS>(text, a, cur, edge, tab, K, po, queue_off);
        } else if (POS) {
            uint64_t mm;
            smh_keyb_lane_slow(text, n, a, tab, K, &mm);
            cnt += smh_append_bits(mm, a, *po);
        } else {
            cnt += smh_keyb_lane_slow(text, n, a, tab, K);
        }
        k = S.take(n_chunks);
    }
    return cnt;
}

#endif
