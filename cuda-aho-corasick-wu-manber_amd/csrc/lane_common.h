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

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
#define SMH_LANE __device__ __forceinline__
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
#define SMH_WAVE_ANY(x) (x)
struct smh_u32x4 { uint32_t v[4]; };
SMH_LANE smh_u32x4 smh_load16(const uint8_t *p)
{
    smh_u32x4 r;
    memcpy(r.v, p, 16);
    return r;
}
#endif

SMH_LANE uint32_t smh_byte_of(uint32_t word, int k) { return (word >> (8 * k)) & 0xFFu; }

#endif
