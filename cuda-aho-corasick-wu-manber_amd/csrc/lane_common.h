/*
 * csrc/lane_common.h -- per-lane building blocks shared by the gfx950 kernels.
 *
 * The scan kernels are written as "what one lane does with its 64-byte text
 * segment"; the __global__ wrappers in ac_kernels.hip / wm_kernels.hip add the
 * LDS staging, the grid-stride loop over wave-chunks and the wave reduction.
 * Keeping the lane bodies free of HIP builtins (apart from the few macros
 * below) lets tests/emu/ compile the very same code with g++ and drive it lane
 * by lane on the CPU, which is how tiling / halo / tail bugs are caught in the
 * GPU-less authoring container.  The emulation is a test harness only; it is
 * not linked into libsmatcher_hip.so.
 *
 * Text geometry (DESIGN.md "Text tiling"):
 *   segment   = 64 consecutive bytes at a 64-byte aligned offset, owned by one lane
 *   wave-chunk= 64 lanes x NCH segments = NCH * 4 KiB of contiguous text
 *   AC  : a lane owns the match START positions inside its segment and reads
 *         m-1 bytes past it (post-halo, HC 16-byte pieces)
 *   WM  : a lane owns the match END columns inside its segment and reads
 *         16*HC >= m-1 bytes before it (pre-halo)
 * Every 16-byte piece is loaded with one global_load_dwordx4; the four loads a
 * wave issues for one segment row touch the same 32 cache lines, so HBM sees
 * each line once.
 */
#ifndef SMH_LANE_COMMON_H
#define SMH_LANE_COMMON_H

#include <stdint.h>
#include <string.h>

#define SMH_SEG 64u /* bytes per lane segment */
/* Register prefetch of a wave's NEXT chunk while the current one is scanned (a second set of text registers, copied over
 * after the scan).  Off: with the table in LDS a CU runs 16 waves, which cover a chunk's load latency by themselves, and the
 * copy (16-20 v_mov per chunk, 7 % of the stride-2 automaton kernel's VALU work) and the registers cost more than the
 * prefetch hides -- stride-2 automaton m = 8: 0.184 -> 0.170 ms/GiB, stride-1 8000 patterns 0.310 -> 0.285, K = 10
 * 0.187 -> 0.171 (gpurun_out/r02_az; profiles/r02_h/notes).  -DSMH_PREFETCH=1 builds the round-2 kernels up to r02_g. */
#ifndef SMH_HYB_COMPACT0
#define SMH_HYB_COMPACT0 0x8000u /* hybrid stride-2 image: id of the first item slot of the compact part (== smh_internal.h; ac_host.c hyb_build, ac_lane.h) */
#endif
#ifndef SMH_PREFETCH
#define SMH_PREFETCH 0
#endif
#ifndef SMH_REGV_MAX_PER_CHUNK
#define SMH_REGV_MAX_PER_CHUNK 8.0 /* == smh_internal.h: surviving columns per 4 KiB wave-chunk up to which the pair-gram kernels verify in registers */
#endif
/* (8000 x 16 at 4.4 per chunk: in registers 10 % faster than staged; 0.2 per chunk and below: 0-2.5 % faster; between 0.5
 * and 3 per chunk against the staged kernel that also drains sparse chunks from HBM: -3 % .. +2 %, within the noise of the
 * interleaved A/Bs -- profiles/r03_experiments/regv_ab_threshold.log, gram_forms_ab.log; 20 per chunk: 12 % slower) */
#ifndef SMH_L2_MIN_PER_CHUNK
#define SMH_L2_MIN_PER_CHUNK 0.02 /* == smh_internal.h: surviving columns per wave-chunk from which the DNA forms verify through the windows-from-L2 pipeline */
#endif
#ifndef SMH_L2_MIN_PER_CHUNK_REGV
#define SMH_L2_MIN_PER_CHUNK_REGV 0.25 /* (both headers) ... for the forms that can verify in registers (pair form, two-column 8-grams): below it -- the headline sets, 0.003-0.03 per chunk -- the in-register instance is 2 % faster (no pipeline to move along) */
#endif
#ifndef SMH_L2_DNA_MAX_PER_CHUNK
#define SMH_L2_DNA_MAX_PER_CHUNK 40.0 /* (both headers) ... up to here: at 46 per chunk the staged verify measured 3 % faster again */
#endif
#ifndef SMH_REGV_WANTED
#define SMH_REGV_WANTED(per_chunk) ((per_chunk) <= SMH_REGV_MAX_PER_CHUNK)
#endif

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
#define SMH_LANE __device__ __forceinline__
#define SMH_MEMBER __device__ __forceinline__
#define SMH_WAVE_ANY(x) (__any((int)(x)) != 0)
struct smh_u32x4 { uint32_t v[4]; };
SMH_LANE smh_u32x4 smh_load16(const uint8_t *p)
{
    const uint4 t = *reinterpret_cast<const uint4 *>(p);
    smh_u32x4 r;
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    return r;
}
#else
#define SMH_LANE static inline
#define SMH_MEMBER inline
#define SMH_WAVE_ANY(x) (x)
struct smh_u32x4 { uint32_t v[4]; };
SMH_LANE smh_u32x4 smh_load16(const uint8_t *p)
{
    smh_u32x4 r;
    memcpy(r.v, p, 16);
    return r;
}
#endif

/*
 * Dword `q` of the text that FOLLOWS a lane's 64-byte segment (its post-halo), without loading
 * it again: those bytes are the first bytes of the next lane's segment, already sitting in that
 * lane's registers, so they are pulled across with one DPP move (wave_shl:1 = "lane i reads lane
 * i+1").  Lane 63 has no right-hand neighbour in the wave; it receives `edge`, which the caller
 * sets to the wave-uniform value that follows the wave's last segment.  Re-loading the halo from
 * memory instead (a 16-byte load per lane at +64) makes every wave touch all 32 cache lines of
 * its chunk a second time, long after the first pass -- measured as 1.9x HBM read traffic.
 * Must be called with all 64 lanes active.  The CPU emulation reads the same bytes from memory.
 */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE uint32_t smh_next_lane_word(uint32_t mine, uint32_t edge, const uint8_t *, uint64_t)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)mine, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
}
SMH_LANE uint32_t smh_first_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
SMH_LANE uint64_t smh_uniform64(uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
#else
SMH_LANE uint32_t smh_next_lane_word(uint32_t, uint32_t, const uint8_t *text, uint64_t byte_offset)
{
    uint32_t v;
    memcpy(&v, text + byte_offset, 4);
    return v;
}
SMH_LANE uint32_t smh_first_lane(uint32_t v) { return v; }
SMH_LANE uint64_t smh_uniform64(uint64_t v) { return v; }
#endif

/*
 * Wave-chunk scheduling.  Static dealing (chunk k to wave k mod nwaves) gives every wave the same
 * amount of text, but the waves of a CU do not run at the same speed -- the four waves of a SIMD share
 * its issue port oldest-first -- and a launch ended 20-25 % after its MEDIAN wave had finished
 * (tools/wavetrace.py, profiles/r02_*).  So the chunks are dealt to WORKGROUPS round-robin (chunk j * nwg +
 * wg belongs to workgroup wg: the chip still streams one contiguous window of text) and, inside a
 * workgroup, taken by whichever wave is free next from a counter in LDS (one ds_add_rtn per chunk, by
 * lane 0).  take() returns a wave-uniform chunk index; indices >= the chunk count end the wave's loop.
 * The CPU emulation deals statically, as before.
 */
/* Order: the LAST chunk and chunk 0 are handed out first.  They are the ones that take the bounds-checked
 * per-lane path (no text in front of chunk 0; no halo behind the last chunk), which is some 30 us for one
 * 4 KiB chunk: taken last, that path alone was a 30 us tail on a 200 us launch. */
SMH_LANE uint64_t smh_sched_order(uint64_t idx, uint64_t n_chunks)
{
    if (idx >= n_chunks) return n_chunks; /* exhausted */
    return idx == 0 ? n_chunks - 1 : idx - 1;
}
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
struct smh_chunk_sched {
    uint32_t ctr_off; /* LDS byte offset of the workgroup's counter (zeroed before the first take) */
    uint32_t nwg, wg;
    SMH_MEMBER uint64_t take(uint64_t n_chunks) const
    {
        uint32_t j = 0;
        if ((threadIdx.x & 63u) == 0)
            j = __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) uint32_t *>(ctr_off), 1u,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
        return smh_sched_order((uint64_t)j * nwg + wg, n_chunks);
    }
};
/* place the counter behind `lds_used` bytes of dynamic LDS (the caller allocates SMH_SCHED_LDS more) and
 * zero it; the caller's next __syncthreads() publishes it */
#define SMH_SCHED_LDS 32u /* the counter (4 bytes); behind it, at +8 / +16 / +24: the launch's start time, stats block and tag (smh_stats.h) */
SMH_LANE smh_chunk_sched smh_sched_init(unsigned char *lds, uint32_t lds_used)
{
    const uint32_t off = (lds_used + 15u) & ~15u;
    if (threadIdx.x == 0) *reinterpret_cast<uint32_t *>(lds + off) = 0u;
    return smh_chunk_sched{off, gridDim.x, blockIdx.x};
}
#else
struct smh_chunk_sched {
    mutable uint64_t next;
    uint64_t step;
    SMH_MEMBER uint64_t take(uint64_t n_chunks) const
    {
        const uint64_t k = next;
        next += step;
        return smh_sched_order(k, n_chunks);
    }
};
/* static dealing for (wave, nwaves): what the CPU emulation uses */
SMH_LANE smh_chunk_sched smh_sched_static(uint64_t wave, uint64_t nwaves) { return smh_chunk_sched{wave, nwaves}; }
#endif

/*
 * Instruction-level helpers.  The scan kernels are VALU-issue bound (every VALU op holds a SIMD's
 * issue port for 4 cycles: profiles/r01_v2_dpp_prefetch), so the inner loops are written as the
 * exact op sequences we want and these helpers pin the instruction choice:
 *   smh_bfe      -> v_bfe_u32
 *   smh_lds_u16  -> ds_read_u16 from a BYTE OFFSET inside the workgroup's LDS allocation.  The
 *                   tables are staged at LDS offset 0 (the kernels use only dynamic LDS), so the
 *                   offset is the address and no base has to be added per lookup.
 */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE uint32_t smh_bfe(uint32_t x, uint32_t off, uint32_t width) { return __builtin_amdgcn_ubfe(x, off, width); }
/* min(wave-uniform bound, v) as ONE v_min_u32 (written as a ternary next to a compare of the same operands
 * the compiler emits v_cmp + v_mov + v_cndmask) */
SMH_LANE uint32_t smh_umin_uniform(uint32_t bound, uint32_t v)
{
    uint32_t r;
    asm("v_min_u32_e32 %0, %1, %2" : "=v"(r) : "s"(bound), "v"(v));
    return r;
}
#define SMH_UNLIKELY(x) __builtin_expect(!!(x), 0)
/* acc + popcount(x) as ONE v_bcnt_u32_b32 (the compiler splits it into v_bcnt + a shared v_add3) */
SMH_LANE uint32_t smh_popc_add(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
SMH_LANE uint32_t smh_lds_u16(const void *, uint32_t byte_off)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>(byte_off);
}
SMH_LANE uint32_t smh_lds_u32(const void *, uint32_t byte_off)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t *>(byte_off);
}
#else
SMH_LANE uint32_t smh_bfe(uint32_t x, uint32_t off, uint32_t width) { return (x >> off) & ((1u << width) - 1u); }
SMH_LANE uint32_t smh_umin_uniform(uint32_t bound, uint32_t v) { return v < bound ? v : bound; }
#define SMH_UNLIKELY(x) (x)
SMH_LANE uint32_t smh_popc_add(uint32_t x, uint32_t acc) { return acc + (uint32_t)__builtin_popcount(x); }
SMH_LANE uint32_t smh_lds_u16(const void *base, uint32_t byte_off)
{
    uint16_t v;
    memcpy(&v, (const uint8_t *)base + byte_off, 2);
    return v;
}
SMH_LANE uint32_t smh_lds_u32(const void *base, uint32_t byte_off)
{
    uint32_t v;
    memcpy(&v, (const uint8_t *)base + byte_off, 4);
    return v;
}
#endif

/*
 * Wavefront-level compaction for position output: every lane announces how many entries it will
 * append; the wave computes an exclusive prefix sum over its 64 lanes, ONE lane advances the global
 * cursor by the wave total with a single atomic, and each lane gets the slot where its own entries
 * start.  Must be called by all 64 lanes.  The CPU emulation appends lane by lane.
 */
#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
SMH_LANE uint64_t smh_wave_reserve(uint64_t *cursor, uint32_t mine)
{
    uint32_t incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off, 64);
        if ((threadIdx.x & 63u) >= (uint32_t)off) incl += up;
    }
    const uint32_t total = __shfl(incl, 63, 64);
    uint64_t base = 0;
    if ((threadIdx.x & 63u) == 0 && total) base = atomicAdd((unsigned long long *)cursor, (unsigned long long)total);
    base = ((uint64_t)__shfl((uint32_t)(base >> 32), 0, 64) << 32) | __shfl((uint32_t)base, 0, 64);
    return base + (incl - mine);
}
#else
SMH_LANE uint64_t smh_wave_reserve(uint64_t *cursor, uint32_t mine)
{
    const uint64_t base = *cursor;
    *cursor += mine;
    return base;
}
#endif

/* where match END columns go in positions mode (smh_ac_positions): a device array, its capacity and
 * the device cursor that counts every match (entries past the capacity are dropped but counted) */
struct smh_pos_out {
    uint64_t *out;
    uint64_t capacity;
    uint64_t *cursor;
};

/* wave-level append of the END columns base + (bit index) of the set bits of `mask`: one vote, one
 * prefix sum, one atomic on the cursor per wave (smh_wave_reserve).  All 64 lanes must call it. */
SMH_LANE uint32_t smh_append_bits(uint64_t mask, uint64_t base, const smh_pos_out &po)
{
    if (!SMH_WAVE_ANY(mask != 0)) return 0;
    const uint32_t mine = (uint32_t)__builtin_popcountll(mask);
    uint64_t slot = smh_wave_reserve(po.cursor, mine);
    while (mask) {
        const int b = __builtin_ctzll(mask);
        mask &= mask - 1;
        if (slot < po.capacity) po.out[slot] = base + (uint64_t)b;
        ++slot;
    }
    return mine;
}

/* two masks, two bases, ONE reservation (a segment's own columns and its halo): the cursor atomic is
 * what bounds dense outputs, so it is spent once per wave and segment */
SMH_LANE uint32_t smh_append_bits2(uint64_t mask_a, uint64_t base_a, uint64_t mask_b, uint64_t base_b, const smh_pos_out &po)
{
    if (!SMH_WAVE_ANY((mask_a | mask_b) != 0)) return 0;
    const uint32_t mine = (uint32_t)(__builtin_popcountll(mask_a) + __builtin_popcountll(mask_b));
    uint64_t slot = smh_wave_reserve(po.cursor, mine);
    while (mask_a) {
        const int b = __builtin_ctzll(mask_a);
        mask_a &= mask_a - 1;
        if (slot < po.capacity) po.out[slot] = base_a + (uint64_t)b;
        ++slot;
    }
    while (mask_b) {
        const int b = __builtin_ctzll(mask_b);
        mask_b &= mask_b - 1;
        if (slot < po.capacity) po.out[slot] = base_b + (uint64_t)b;
        ++slot;
    }
    return mine;
}

SMH_LANE uint32_t smh_byte_of(uint32_t word, int k) { return (word >> (8 * k)) & 0xFFu; }

#endif
