#!/bin/bash
# tools/tsan_adapt.sh -- csrc/smh_adapt.h (the adaptive engine's host state and policy) under ThreadSanitizer on the CPU.
# With the mutex: no report.  Without it ("nolock"): ThreadSanitizer must report, i.e. the harness sees what the mutex guards.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
g++ -std=c++17 -O1 -g -fsanitize=thread -I"$ROOT/cuda-aho-corasick-wu-manber_amd/csrc" "$ROOT/tools/tsan_adapt.cpp" -o /tmp/tsan_adapt -pthread
/tmp/tsan_adapt 2>&1 | tee /tmp/tsan_adapt.log
if grep -q "ThreadSanitizer" /tmp/tsan_adapt.log; then echo "TSAN REPORTS with the mutex held"; exit 1; fi
if /tmp/tsan_adapt nolock > /tmp/tsan_adapt_nolock.log 2>&1 && ! grep -q "ThreadSanitizer" /tmp/tsan_adapt_nolock.log; then
    echo "note: the unlocked run raised no report this time (races are timing-dependent)"
else
    echo "unlocked control run: ThreadSanitizer reports $(grep -c 'WARNING: ThreadSanitizer' /tmp/tsan_adapt_nolock.log) race(s), as expected"
fi
echo "tsan: clean with the mutex"
