/*
 * csrc/smh_adapt.h -- the adaptive engine's host-side state and policy (round 4; moved out of smh_runtime.hip in round 5 so that
 * tools/tsan_adapt.cpp can build it for the CPU with -fsanitize=thread): which engine a handle runs next, from the records its
 * launches publish (smh_stats.h).  No HIP call in here: the device blocks and the pinned host records are allocated by
 * smh_runtime.hip (adapt_get), which also holds smh_adapt_dev.mu around every call below.
 */
#ifndef SMH_ADAPT_H
#define SMH_ADAPT_H

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "smh_internal.h"
#include "smh_stats.h"

/* ------------------------------------------------------------------ adaptive engine (round 4; smh_stats.h)
 * A handle that holds several engines -- an automaton with a depth-cut or hybrid plan, the suffix-filter kernels over
 * the same patterns (smh_ac.flex_wm / smh_wm.flex_ac), the plain stride-1 automaton (smh_ac.flat_ac) -- starts with the
 * one its compile estimated fastest on random text and then follows the launches' own reports: every count launch of
 * 32 MiB or more publishes its duration (device clock, one workgroup's prologue to its last chunk) and the number of
 * columns it had to verify (a sample of eight workgroups, scaled: smh_stats.h).  Before the NEXT launch the host compares, per GiB, the running engine's measured time
 * (the better of its last two reports: an engine's first launch on a device runs cold) with the best of the others
 * (measured on this text, else the compile's estimate) and switches when that is clearly better -- after the first
 * report when a launch of 256 MiB or more ran four times slower than estimated, else after the second.  What was measured of an
 * engine that is not running is forgotten, so that it is tried again, when the running engine's events per 4 KiB move by
 * a factor of two (another kind of text) and after 32 reports of the others -- 64, 128, ... 4096 when it keeps losing.  The
 * filter kernels' verify mode (in registers / staged) follows the measured survivors per chunk the same way.  Not every
 * launch reports (adapt_arg: a report costs its launch 2-4 us).  Nothing here synchronises: a launch that has not finished has simply not reported yet.  SMH_ADAPT=0 in the environment
 * (read once) turns all of it off; a forced engine or plan is never overridden. */
struct smh_adapt_dev {
    int device;
    smh_adapt_dev *next;
    std::mutex *mu;        /* round 5: everything below is read and written under it -- tuned scans of one handle may come from several host threads */
    smh_scan_stats *d_stats; /* SMH_STATS_SLOTS blocks: a reporting launch takes the next one (smh_stats.h) */
    unsigned long long *h_rec; /* pinned host records the last workgroup of a launch writes: SMH_STATS_SLOTS x SMH_STATS_HOST_WORDS */
    uint64_t *d_scratch;   /* a count nobody reads (adapt_first_look's probes of the other engines) */
    unsigned int seen[SMH_STATS_SLOTS];
    unsigned int slot_nonce[SMH_STATS_SLOTS]; /* the nonce of the launch that holds the slot */
    unsigned int next_slot, next_nonce;
    void *last_stream;     /* of the newest tuned launch */
    int have_stream;
    int multi_stream;      /* the handle has been launched on more than one stream on this device: every launch now records order_ev behind
                            * itself and a launch on another stream than the previous one waits for it (smh_runtime.hip adapt_order_*) */
    void *order_ev;        /* hipEvent_t */
    int unordered;         /* the launch being issued could not be ordered behind the previous one (stream capture in progress) */
    int engine;            /* the kernels that run next; -1 before the first launch */
    int fresh;             /* a report of the running engine arrived since the last decision */
    double last[SMH_ENGINES][2]; /* per engine: ms per GiB of its last two reports, [0] the newer */
    int n[SMH_ENGINES];    /* reports held (0..2); 0 = not measured on this text */
    uint32_t age[SMH_ENGINES]; /* reports of other engines since */
    uint32_t keep[SMH_ENGINES]; /* reports of other engines after which the engine's measurement is forgotten: 32, doubling
                                 * every time it is (an engine that keeps losing is tried ever more rarely), back to 32 when it wins */
    double sig[SMH_ENGINES]; /* events per 4 KiB at the engine's last report */
    unsigned long long last_bytes; /* text length of the newest report */
    double ref_sig;
    int ref_valid;
    uint32_t reports, flips;
    uint32_t launches;     /* tuned count launches of the handle on this device */
    double mode_density;   /* survivors per column handed to the gram launcher (< 0: the compile's estimate so far) */
    double slow;           /* the most a text-dependent engine has run over its estimate on this kind of text (>= 1) */
    int tried[SMH_ENGINES]; /* the engine has reported on this kind of text */
};

/* engines whose rate does not depend on the text: their compile-time estimate holds on any text.  (The window-hash engine's filter
 * rate does not; it verifies the true matches, two round trips each -- its estimate holds within a small factor while matches are
 * a few per cent of the columns, where a q-gram filter's is off by the 5-30 x the policy otherwise assumes.) */
static bool engine_text_independent(int e) { return e == SMH_ENGINE_AC_FLAT || e == SMH_ENGINE_KEYS || e == SMH_ENGINE_HASH; }

static bool adapt_enabled()
{
    static const int on = [] { const char *e = getenv("SMH_ADAPT"); return e && atoi(e) == 0 ? 0 : 1; }();
    return on != 0;
}

static double adapt_ms(const smh_adapt_dev *A, int e)
{
    if (A->n[e] == 0) return 0.0;
    return A->n[e] == 1 || A->last[e][0] < A->last[e][1] ? A->last[e][0] : A->last[e][1];
}

#define SMH_ADAPT_MIN_BYTES (32ull << 20) /* smaller launches are mostly table staging and tail: not a rate */
#define SMH_ADAPT_FIXED_TICKS 400.0        /* 4 us of every launch are table staging and the last wave's tail whatever the text's length: taken off before a duration becomes a rate */
static void adapt_poll_slot(smh_adapt_dev *A, unsigned int slot)
{
    volatile unsigned long long *h = A->h_rec + slot * SMH_STATS_HOST_WORDS;
    const unsigned int seq = (unsigned int)h[0];
    if (seq == A->seen[slot]) return;
    const unsigned long long ev = h[1], ticks = h[2], bytes = h[3], tagw = h[4], seq2 = h[5], sum = h[6];
    /* the record is written without a fence between data and flag: it validates itself */
    if ((unsigned int)seq2 != seq || (unsigned int)h[0] != seq || sum != (ev ^ ticks ^ bytes ^ tagw ^ (unsigned long long)seq)) return;
    A->seen[slot] = seq;
    /* (a record is one launch's whatever launch holds the slot by now -- the device publishes only tickets whose workgroups all
     * carried the same nonce -- so a caller that queues launches far ahead of the device still gets its reports, late) */
    const unsigned int tag = (unsigned int)tagw;
    const int e = (int)(tag & 0xFFu);
    if (bytes < SMH_ADAPT_MIN_BYTES || ticks == 0 || e >= SMH_ENGINES) return;
    A->sig[e] = (double)ev * 4096.0 / (double)bytes;
    if (tag & (1u << 20)) return; /* launched beside another launch of the handle on another stream: its events count, its duration does not */
    A->last[e][1] = A->last[e][0];
    double t = (double)ticks - SMH_ADAPT_FIXED_TICKS;
    if (t < 0.25 * (double)ticks) t = 0.25 * (double)ticks;
    const double launches = (double)(((tag >> 8) & 0xFFu) ? ((tag >> 8) & 0xFFu) : 1u); /* the first of that many equal launches reported (ac_flat_launch) */
    A->last[e][0] = launches * t * 1e-5 * (double)(1ull << 30) / (double)bytes; /* 100 MHz ticks -> ms per GiB */
    A->last_bytes = bytes;
    if (A->n[e] < 2) ++A->n[e];
    A->tried[e] = 1;
    A->age[e] = 0;
    for (int o = 0; o < SMH_ENGINES; ++o)
        if (o != e) ++A->age[o];
    ++A->reports;
    if (e == A->engine) A->fresh = 1;
}
static void adapt_poll(smh_adapt_dev *A)
{
    for (unsigned int slot = 0; slot < SMH_STATS_SLOTS; ++slot) adapt_poll_slot(A, slot);
}

/* which engine runs the next launch; est[e] = the compile's estimate in ms per GiB, <= 0: the handle has no such engine */
static int adapt_choose(smh_adapt_dev *A, const double est[SMH_ENGINES], int initial)
{
    if (A->engine < 0) { A->engine = initial; return initial; }
    if (!A->fresh) return A->engine;
    A->fresh = 0;
    const int cur = A->engine;
    const double c_cur = adapt_ms(A, cur);
    if (c_cur <= 0) return cur;
    /* a first report runs cold: wait for the second -- unless a launch of 256 MiB or more took four times what was estimated */
    if (A->n[cur] < 2 && !(est[cur] > 0 && c_cur > 4.0 * est[cur] && A->last_bytes >= (256ull << 20))) return cur;
    if (!A->ref_valid) {
        A->ref_sig = A->sig[cur];
        A->ref_valid = 1;
    } else {
        const double a = A->sig[cur], b = A->ref_sig;
        if (fabs(a - b) > 0.05 && (a > 2.0 * b || b > 2.0 * a)) { /* another kind of text: what the others did on the old one says nothing */
            for (int o = 0; o < SMH_ENGINES; ++o)
                if (o != cur) A->n[o] = 0, A->tried[o] = 0;
            A->ref_sig = a;
            A->slow = 1.0;
        }
    }
    /* An estimate is a rate on random text.  Text that slows one text-dependent engine (survivors to verify, lanes deep in
     * compact rows) slows the other for the same reason -- measured on the non-uniform corpora 5-30 x for the filter
     * kernels where the hybrid image ran 7-75 x over -- so an engine of that kind that has NOT run on this text yet is
     * expected to be off by the factor the running one is; the plain stride-1 parts are not (their estimate holds on any
     * text).  An engine whose measurement was merely forgotten (below) is re-tried at its plain estimate. */
    if (!engine_text_independent(cur) && est[cur] > 0 && c_cur / est[cur] > A->slow) A->slow = c_cur / est[cur];
    int best = -1;
    double c_best = 0, m_best = 1.0;
    for (int o = 0; o < SMH_ENGINES; ++o) {
        if (o == cur || est[o] <= 0) continue;
        if (A->keep[o] == 0) A->keep[o] = 32u;
        if (A->n[o] > 0 && A->age[o] >= A->keep[o]) { /* the race is re-run now and then: the text may have changed in a way the running engine's events do not show */
            A->n[o] = 0;
            if (A->keep[o] < 4096u) A->keep[o] *= 2u;
        }
        double c = A->n[o] > 0 ? adapt_ms(A, o) : est[o];
        if (A->n[o] == 0 && !engine_text_independent(o) && !A->tried[o] && A->slow > 2.0) c *= A->slow;
        const double margin = A->n[o] > 0 ? 1.03 : 1.08;
        if (best < 0 || c * margin < c_best * m_best) { best = o; c_best = c; m_best = margin; }
    }
    if (best >= 0 && c_best * m_best < c_cur) {
        A->engine = best;
        A->n[best] = 0; /* a fresh series for the engine that takes over */
        if (c_best * 1.5 < c_cur) A->keep[best] = 32u;
        A->ref_valid = 0;
        ++A->flips;
    }
    return A->engine;
}

/* surviving columns per text column the gram launcher should plan its verify mode for: the compile's estimate until
 * the filter kernels have reported from this text, then what they measured -- replaced only when it moves by more than a
 * quarter, so that a rate near one of the launcher's thresholds does not flip the kernel instance from launch to launch */
static float adapt_density(smh_adapt_dev *A, const struct smh_wm *wm)
{
    if (!A) return (float)wm->gram_density;
    if (A->mode_density < 0) A->mode_density = wm->gram_density;
    if (A->n[SMH_ALGO_WM] > 0) {
        const double meas = A->sig[SMH_ALGO_WM] / 4096.0, old = A->mode_density;
        if (fabs(meas - old) > 0.25 * (meas > old ? meas : old)) A->mode_density = meas;
    }
    return (float)A->mode_density;
}

/* A report costs its launch 2-4 us (the reporting workgroups' atomics, the record's trip to host memory before the kernel
 * may end: measured 1-3 % on the 175 us headline scans), so not every launch reports: the first four of a handle on a
 * device, every launch while the running engine's series is incomplete (a decision is pending), then every eighth. */
static smh_stats_arg adapt_slot(smh_adapt_dev *A, uint64_t n, int engine, bool unreliable)
{
    smh_stats_arg sa = {};
    const unsigned int slot = A->next_slot++ % SMH_STATS_SLOTS;
    A->next_nonce = A->next_nonce % 4095u + 1u; /* 1..4095 */
    A->slot_nonce[slot] = A->next_nonce;
    sa.st = A->d_stats + slot;
    sa.bytes = n;
    sa.tag = (unsigned int)engine | slot << 16 | (unreliable ? 1u << 20 : 0u);
    sa.nonce = A->next_nonce;
    return sa;
}
/* (launches under SMH_ADAPT_MIN_BYTES never report -- adapt_poll would discard the record, and a handle that only ever scans
 * short texts would pay the report on every launch for a series that never completes) */
static smh_stats_arg adapt_arg(smh_adapt_dev *A, uint64_t n, int engine, void *stream)
{
    smh_stats_arg sa = {};
    if (!A) return sa;
    /* (launches of one handle are ordered on the device also across streams -- smh_runtime.hip adapt_order_before -- so a launch's
     * duration is its own; inside a stream capture that ordering is not applied and the duration of a launch on another stream than
     * the previous one is not used) */
    (void)stream;
    const bool unreliable = A->unordered != 0;
    const uint32_t k = A->launches++;
    if (n < SMH_ADAPT_MIN_BYTES) return sa;
    /* (an unreliable launch's record is never counted into A->n, so the "series incomplete" rule must not apply to it: a handle that
     * is only ever launched unordered -- inside captures -- would otherwise pay the report on every launch, for ever) */
    const bool series_open = !unreliable && engine >= 0 && engine < SMH_ENGINES && A->n[engine] < 2;
    if (k < 4u || series_open || (k & 7u) == 0u) sa = adapt_slot(A, n, engine, unreliable);
    return sa;
}

static void adapt_report(const smh_adapt_dev *A, int adaptive, int engine_static, const double est[SMH_ENGINES], smh_adapt_info *out)
{
    const uint32_t size = out->struct_size;
    memset(out, 0, sizeof *out);
    out->struct_size = size;
    out->adaptive = (uint32_t)adaptive;
    out->engine = (uint32_t)(adaptive && A && A->engine >= 0 ? A->engine : engine_static);
    for (int e = 0; e < SMH_ENGINES; ++e) out->est_ms_per_gib[e] = est[e];
    out->verify_density = -1.0;
    if (!A) return;
    out->flips = A->flips;
    out->reports = A->reports;
    for (int e = 0; e < SMH_ENGINES; ++e) { out->ms_per_gib[e] = adapt_ms(A, e); out->events_per_4k[e] = A->sig[e]; }
    out->verify_density = A->mode_density;
}

#endif
