/*
 * csrc/smh_stats.h -- what a scan launch tells the host about itself (round 4).
 *
 * The engine that serves an entry point is chosen at compile time from rates measured on pseudo-random text
 * (ac_host.c, wm_host.c).  A filter engine's speed, though, is a property of the TEXT: a column that survives the
 * filter costs a window hash and a table probe, and on repeat-rich text (the reference's E.coli / swiss-prot,
 * main.c:39-109) survivors are not rare.  So the filter kernels and the depth-cut automaton kernels count their
 * surviving columns / candidates (wave-uniform scalar adds beside the compaction they do anyway), every workgroup
 * adds its sum and its start / end time (the 100 MHz s_memrealtime counter) to a block in device memory, and the
 * LAST workgroup of the launch publishes {events, ticks from the first start to the last end, bytes, tag} to a
 * record in pinned host memory and clears the block.  The host reads the record before its NEXT launch of the handle
 * -- no synchronisation, a launch that has not finished simply has not reported yet -- and may then run the other
 * engine or another verify mode (smh_runtime.hip "adaptive engine").
 */
#ifndef SMH_STATS_H
#define SMH_STATS_H

#include <stdint.h>

struct smh_scan_stats { /* device memory, one per (handle, device); zero except t_min = ~0 */
    unsigned long long t_min, t_max, events;
    unsigned int done, seq;
    unsigned long long *host; /* SMH_STATS_HOST_WORDS words of pinned host memory: seq, events, ticks, bytes, tag */
    unsigned long long pad[3];
};
#define SMH_STATS_HOST_WORDS 8

struct smh_stats_arg { /* kernel argument; st == NULL: the launch reports nothing */
    smh_scan_stats *st;
    unsigned long long bytes;
    unsigned int tag;
    unsigned int pad;
};

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
__device__ __forceinline__ uint64_t smh_stats_now(const smh_stats_arg &A) { return A.st ? __builtin_amdgcn_s_memrealtime() : 0ull; }

/* one thread per workgroup, after the workgroup's last text access */
__device__ __forceinline__ void smh_stats_commit(const smh_stats_arg &A, uint64_t t_start, uint64_t events)
{
    smh_scan_stats *st = A.st;
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    if (events) atomicAdd(&st->events, (unsigned long long)events);
    atomicMin(&st->t_min, (unsigned long long)t_start);
    atomicMax(&st->t_max, t_end);
    __threadfence();
    const unsigned int ticket = atomicAdd(&st->done, 1u);
    if (ticket + 1u == gridDim.x) {
        __threadfence();
        const unsigned long long e = atomicExch(&st->events, 0ull), t0 = atomicExch(&st->t_min, ~0ull),
                                 t1 = atomicExch(&st->t_max, 0ull);
        atomicExch(&st->done, 0u);
        const unsigned int seq = atomicAdd(&st->seq, 1u) + 1u;
        volatile unsigned long long *h = st->host;
        h[1] = e;
        h[2] = t1 - t0;
        h[3] = A.bytes;
        h[4] = A.tag;
        __threadfence_system();
        h[0] = seq;
    }
}

/* the workgroup's match count into *count with ONE atomic (as smh_block_add) and, when the launch reports, its events and
 * times into the stats block.  `wave_events` is wave-uniform.  `lds` may be the table region: the first barrier makes sure
 * every wave is done reading it. */
__device__ __forceinline__ void smh_block_finish(uint32_t cnt, uint64_t *count, unsigned char *lds, const smh_stats_arg &A,
                                                 uint64_t t_start, uint32_t wave_events)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
    __syncthreads();
    uint32_t *part = reinterpret_cast<uint32_t *>(lds);
    if ((threadIdx.x & 63u) == 0) {
        part[threadIdx.x >> 6] = cnt;
        part[16u + (threadIdx.x >> 6)] = wave_events;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const bool in = threadIdx.x < (blockDim.x >> 6);
        uint64_t v = in ? part[threadIdx.x] : 0u;
        uint64_t ev = in ? part[16u + threadIdx.x] : 0u;
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            v += __shfl_down(v, off, 64);
            ev += __shfl_down(ev, off, 64);
        }
        if (threadIdx.x == 0) {
            if (v && count) atomicAdd((unsigned long long *)count, (unsigned long long)v);
            if (A.st) smh_stats_commit(A, t_start, ev);
        }
    }
}
#endif

#endif
