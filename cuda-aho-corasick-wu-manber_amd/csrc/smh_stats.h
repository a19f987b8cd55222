/*
 * csrc/smh_stats.h -- what a scan launch tells the host about itself (round 4).
 *
 * The engine that serves an entry point is chosen at compile time from rates measured on pseudo-random text
 * (ac_host.c, wm_host.c).  A filter engine's speed, though, is a property of the TEXT: a column that survives the
 * filter costs a window hash and a table probe, and on repeat-rich text (the reference's E.coli / swiss-prot,
 * main.c:39-109) survivors are not rare.  So the filter kernels and the depth-cut automaton kernels count their
 * surviving columns / candidates (one add per lane and event beside the compaction they do anyway), a sample of the
 * launch's workgroups adds its sum to a block in device memory, and the LAST of them publishes {events scaled to the
 * grid, its own duration on the 100 MHz s_memrealtime counter, bytes, tag} to a record in pinned host memory and clears
 * the block.  The host reads the record before its NEXT launch of the handle
 * -- no synchronisation, a launch that has not finished simply has not reported yet -- and may then run the other
 * engine or another verify mode (smh_runtime.hip "adaptive engine").
 */
#ifndef SMH_STATS_H
#define SMH_STATS_H

#include <stdint.h>

struct smh_scan_stats { /* device memory, SMH_STATS_SLOTS per (handle, device); zero */
    unsigned long long ticket; /* bits 0..15: reporting workgroups done; 16..47: their events; 48..63: the sum of their launches' nonces */
    unsigned int seq, pad0;
    unsigned long long *host; /* SMH_STATS_HOST_WORDS words of pinned host memory: seq, events, ticks, bytes, tag | nonce << 32, seq, checksum */
    unsigned long long pad[5];
};
#define SMH_STATS_HOST_WORDS 8
#define SMH_STATS_SAMPLE 8u /* workgroups of a launch that report (blockIdx.x below this) */
/* Round 5: launches of ONE handle may overlap (two streams, two host threads: the extended API allows it, the reference's
 * globals did not, smatcher.h:71-73).  Every reporting launch takes the next of SMH_STATS_SLOTS blocks (its own ticket, its own
 * host record) and carries a 12-bit nonce that its workgroups add into the ticket's top bits: the workgroup that completes a
 * ticket publishes only when the sum says all of the ticket's workgroups belonged to ITS launch -- more than SMH_STATS_SLOTS
 * reporting launches in flight at once then lose their reports instead of mixing them. */
#define SMH_STATS_SLOTS 4u

struct smh_stats_arg { /* kernel argument; st == NULL: the launch reports nothing */
    smh_scan_stats *st;   /* the launch's slot */
    unsigned long long bytes;
    unsigned int tag;     /* engine | launches that share this report << 8 | slot << 16 | rate unreliable << 20 (smh_runtime.hip adapt_arg) */
    unsigned int nonce;   /* 1..4095 */
};

#if defined(__HIPCC__) && !defined(SMH_HOST_EMU)
/* The scan kernels run at the edge of their scalar register budget (the pair-gram kernel: 106 SGPRs, no spill): a
 * start time, a pointer and a tag kept alive across the chunk loop for the epilogue's sake cost it nine spilled SGPRs.
 * So thread 0 parks them in the 24 spare bytes behind the workgroup's chunk counter in LDS (lane_common.h
 * SMH_SCHED_LDS) in the prologue and takes them back in the epilogue. */
__device__ __forceinline__ void smh_stats_stash(uint32_t ctr_off, const smh_stats_arg &A)
{
    if (threadIdx.x == 0) {
        const bool on = A.st && blockIdx.x < SMH_STATS_SAMPLE;
        const unsigned long long t = on ? __builtin_amdgcn_s_memrealtime() : 0ull;
        *reinterpret_cast<__attribute__((address_space(3))) unsigned long long *>(ctr_off + 8u) = t;
        *reinterpret_cast<__attribute__((address_space(3))) unsigned long long *>(ctr_off + 16u) = on ? (unsigned long long)(uintptr_t)A.st : 0ull;
        *reinterpret_cast<__attribute__((address_space(3))) unsigned int *>(ctr_off + 24u) = A.tag;
        *reinterpret_cast<__attribute__((address_space(3))) unsigned int *>(ctr_off + 28u) = A.nonce;
    }
}

/* One thread of each REPORTING workgroup, after the workgroup's last text access.  Same-address atomics of workgroups that
 * finish together queue up behind each other at ~12 ns apiece (the reason the match count takes one per workgroup, not one
 * per wave): a first version with four atomics in each of the 256 workgroups put 10 us on the end of a 175 us launch.
 * So only the first SMH_STATS_SAMPLE workgroups report -- the chunks are dealt round-robin, every workgroup sees an even
 * sample of the text and they all run until the chunks are gone -- with ONE atomic each: events and a ticket in one
 * 64-bit add.  The last of them publishes the sample's events scaled to the grid and ITS OWN duration (start of its
 * prologue to here) to the host record.  The record carries its sequence number twice and a checksum instead of a
 * system-scope fence between data and flag (a PCIe round trip at the end of the kernel). */
__device__ __forceinline__ void smh_stats_commit(smh_scan_stats *st, unsigned int tag, unsigned int nonce, unsigned long long bytes, uint64_t t_start, uint64_t events)
{
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    const unsigned int reporting = gridDim.x < SMH_STATS_SAMPLE ? gridDim.x : SMH_STATS_SAMPLE;
    const unsigned long long nn = nonce & 0xFFFu;
    const unsigned long long old = atomicAdd(&st->ticket, (nn << 48) | ((unsigned long long)(events & 0xFFFFFFFFull) << 16) | 1ull);
    if ((unsigned int)(old & 0xFFFFu) + 1u == reporting) {
        atomicExch(&st->ticket, 0ull);
        if ((old >> 48) + nn != nn * reporting) return; /* another launch's workgroups are in this ticket: no report (smh_stats.h SMH_STATS_SLOTS) */
        const unsigned long long e = (((old >> 16) & 0xFFFFFFFFull) + events) * gridDim.x / reporting;
        const unsigned int seq = ++st->seq; /* this thread alone: the ticket was this launch's */
        const unsigned long long ticks = t_end - t_start, tagw = (unsigned long long)tag | (nn << 32);
        volatile unsigned long long *h = st->host;
        h[1] = e;
        h[2] = ticks;
        h[3] = bytes;
        h[4] = tagw;
        h[6] = e ^ ticks ^ bytes ^ tagw ^ (unsigned long long)seq;
        h[5] = seq;
        h[0] = seq;
    }
}

/* the workgroup's match count into *count with ONE atomic (as smh_block_add) and, when the workgroup reports, its events
 * and duration into the stats block.  `wave_events` is the lane's own event count; ctr_off = the chunk counter's LDS offset
 * (where the prologue parked the launch's stats pointer); bytes = the text length.  `lds` may be the table region: the first
 * barrier makes sure every wave is done reading it. */
__device__ __forceinline__ void smh_block_finish(uint32_t cnt, uint64_t *count, unsigned char *lds, uint32_t ctr_off,
                                                 unsigned long long bytes, uint32_t wave_events)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 64);
        wave_events += __shfl_down(wave_events, off, 64);
    }
    unsigned long long t_start = 0, st_bits = 0;
    unsigned int tag = 0, nonce = 0;
    if (threadIdx.x == 0) { /* before the reduction below reuses the front of LDS (a tiny table's counter sits there) */
        t_start = *reinterpret_cast<__attribute__((address_space(3))) unsigned long long *>(ctr_off + 8u);
        st_bits = *reinterpret_cast<__attribute__((address_space(3))) unsigned long long *>(ctr_off + 16u);
        tag = *reinterpret_cast<__attribute__((address_space(3))) unsigned int *>(ctr_off + 24u);
        nonce = *reinterpret_cast<__attribute__((address_space(3))) unsigned int *>(ctr_off + 28u);
    }
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
            if (st_bits) smh_stats_commit(reinterpret_cast<smh_scan_stats *>((uintptr_t)st_bits), tag, nonce, bytes, t_start, ev);
        }
    }
}
#endif

#endif
