/*
 * tools/tsan_adapt.cpp -- the adaptive engine's host-side state machine (csrc/smh_adapt.h) under ThreadSanitizer, on the CPU.
 *
 * Several host threads drive ONE handle's adaptive state the way smh_ac_scan / smh_wm_scan do -- lock, poll the launches'
 * records, choose the engine, take a report slot -- on two "streams"; the launch itself is played by the calling thread, which
 * completes the slot's record the way the last reporting workgroup does (smh_stats.h smh_stats_commit).  A third thread reads the
 * state as smh_*_get_adapt does.  Build and run (tools/tsan_adapt.sh):
 *     g++ -std=c++17 -O1 -g -fsanitize=thread -Icuda-aho-corasick-wu-manber_amd/csrc tools/tsan_adapt.cpp -o /tmp/tsan_adapt -pthread
 *     /tmp/tsan_adapt            -> "ok", exit 0, no ThreadSanitizer report
 *     /tmp/tsan_adapt nolock     -> the same without the mutex: ThreadSanitizer must report (the harness can see what it guards)
 */
#include <stdio.h>
#include <thread>
#include <vector>
#include "smh_adapt.h"

static void complete(const smh_stats_arg &sa, double ms_per_gib, unsigned long long events)
{
    if (!sa.st) return;
    smh_scan_stats *st = sa.st;
    const unsigned int seq = ++st->seq;
    const unsigned long long ticks = (unsigned long long)(ms_per_gib * 1e5 * (double)sa.bytes / (double)(1ull << 30)) + 400ull;
    const unsigned long long tagw = (unsigned long long)sa.tag | ((unsigned long long)(sa.nonce & 0xFFFu) << 32);
    volatile unsigned long long *h = st->host;
    h[1] = events; h[2] = ticks; h[3] = sa.bytes; h[4] = tagw;
    h[6] = events ^ ticks ^ sa.bytes ^ tagw ^ (unsigned long long)seq;
    h[5] = seq; h[0] = seq;
}

int main(int argc, char **argv)
{
    const bool nolock = argc > 1 && !strcmp(argv[1], "nolock");
    smh_adapt_dev A;
    memset((void *)&A, 0, sizeof A);
    A.engine = -1; A.mode_density = -1.0; A.slow = 1.0;
    A.mu = new std::mutex();
    std::vector<unsigned long long> rec(SMH_STATS_SLOTS * SMH_STATS_HOST_WORDS, 0ull);
    std::vector<smh_scan_stats> blocks(SMH_STATS_SLOTS);
    memset(blocks.data(), 0, sizeof(smh_scan_stats) * SMH_STATS_SLOTS);
    A.h_rec = rec.data();
    A.d_stats = blocks.data();
    for (unsigned i = 0; i < SMH_STATS_SLOTS; ++i) blocks[i].host = rec.data() + i * SMH_STATS_HOST_WORDS;
    const double est[SMH_ENGINES] = {0.20, 0.18, 0.55, 0.40, 0.0};
    /* the text turns hostile half way: the filter engine's rate collapses, the text-independent ones hold */
    auto rate = [&](int engine, int step) { return step < 100 ? est[engine] : (engine == SMH_ALGO_AC ? 6.0 : engine == SMH_ALGO_WM ? 3.0 : est[engine]); };
    struct smh_wm wm;
    memset(&wm, 0, sizeof wm);
    wm.gram_density = 0.001;
    int engines_seen[SMH_ENGINES] = {0, 0, 0, 0, 0};
    std::mutex seen_mu;
    auto worker = [&](int id) {
        void *stream = (void *)(uintptr_t)(0x1000 + id);
        for (int step = 0; step < 200; ++step) {
            std::unique_lock<std::mutex> lock(*A.mu, std::defer_lock);
            if (!nolock) lock.lock();
            adapt_poll(&A);
            const int engine = adapt_choose(&A, est, SMH_ALGO_WM);
            const smh_stats_arg sa = adapt_arg(&A, 256ull << 20, engine, stream);
            (void)adapt_density(&A, &wm);
            complete(sa, rate(engine, step), step < 100 ? 10 : 500000);
            if (!nolock) lock.unlock();
            std::lock_guard<std::mutex> g(seen_mu);
            ++engines_seen[engine];
        }
    };
    std::thread reader([&] {
        for (int i = 0; i < 400; ++i) {
            smh_adapt_info out;
            out.struct_size = sizeof out;
            std::unique_lock<std::mutex> lock(*A.mu, std::defer_lock);
            if (!nolock) lock.lock();
            adapt_report(&A, 1, SMH_ALGO_WM, est, &out);
        }
    });
    std::thread t0(worker, 0), t1(worker, 1);
    t0.join(); t1.join(); reader.join();
    printf("launches by engine: automaton %d, filter %d, flat parts %d, key table %d; reports %u, flips %u, engine now %d\n",
           engines_seen[0], engines_seen[1], engines_seen[2], engines_seen[3], A.reports, A.flips, A.engine);
    /* (a launch on another stream than the handle's previous one may have run beside it: its DURATION is not used, smh_adapt.h
     * adapt_arg -- how often that happens above depends on how the two threads interleave.)  The same sequence on ONE stream: */
    smh_adapt_dev B;
    memset((void *)&B, 0, sizeof B);
    B.engine = -1; B.mode_density = -1.0; B.slow = 1.0;
    std::vector<unsigned long long> rec2(SMH_STATS_SLOTS * SMH_STATS_HOST_WORDS, 0ull);
    std::vector<smh_scan_stats> blocks2(SMH_STATS_SLOTS);
    memset(blocks2.data(), 0, sizeof(smh_scan_stats) * SMH_STATS_SLOTS);
    B.h_rec = rec2.data(); B.d_stats = blocks2.data();
    for (unsigned i = 0; i < SMH_STATS_SLOTS; ++i) blocks2[i].host = rec2.data() + i * SMH_STATS_HOST_WORDS;
    for (int step = 0; step < 200; ++step) {
        adapt_poll(&B);
        const int engine = adapt_choose(&B, est, SMH_ALGO_WM);
        complete(adapt_arg(&B, 256ull << 20, engine, (void *)0x1000), rate(engine, step), step < 100 ? 10 : 500000);
    }
    printf("one stream: reports %u, flips %u, engine now %d\n", B.reports, B.flips, B.engine);
    const bool ok = B.engine == SMH_ENGINE_KEYS && B.flips >= 1 && engines_seen[0] + engines_seen[1] + engines_seen[2] + engines_seen[3] == 400;
    printf(ok ? "ok\n" : "UNEXPECTED STATE\n");
    delete A.mu;
    return ok ? 0 : 1;
}
