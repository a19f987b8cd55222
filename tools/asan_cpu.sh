#!/bin/bash
# tools/asan_cpu.sh -- the host code (csrc/*.c), the lane emulator (tests/emu) and the oracle built with
# -fsanitize=address,undefined in a scratch copy of the tree, and the CPU test suite run against that build.
# (Sanitizers run on the CPU build only: the GPU pool refuses GPU AddressSanitizer.)  The HIP translation units
# are compiled as usual; libasan / libubsan come in through LD_PRELOAD.  Tests that need files the copy leaves out
# (profiles/) or that link a C driver without the sanitizer runtime are deselected.
set -e
SRC=$(cd "$(dirname "$0")/.." && pwd)
DST=${1:-/tmp/smh_asan}
rm -rf "$DST" && mkdir -p "$DST"
tar -C "$SRC" --exclude=.git --exclude=gpurun_out --exclude=profiles --exclude='*.so' --exclude=build -cf - . | tar -xf - -C "$DST"
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
make -s -C "$DST/cuda-aho-corasick-wu-manber_amd" CC="gcc $SAN" -j8 libsmatcher_hip.so
# round 6: the -DSMH_TESTING twin as well (its C files through the same CC): the knob-driven tests run under the sanitizer too
make -s -C "$DST/cuda-aho-corasick-wu-manber_amd" CC="gcc $SAN" -j8 ../tests/emu/libsmatcher_hip_testing.so
g++ -O1 $SAN -fPIC -std=c++17 -DSMH_HOST_EMU -I"$DST/cuda-aho-corasick-wu-manber_amd/csrc" -shared \
    -o "$DST/tests/emu/libsmh_emu.so" "$DST/tests/emu/emu_kernels.cpp"
make -s -C "$DST/oracle"
cd "$DST"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
python -m pytest tests -q -m "not gpu" -p no:cacheprovider -n 4 \
    --deselect tests/test_bench_helpers.py --deselect tests/test_multi_device.py --deselect tests/test_c_driver.py --deselect tests/test_concurrent_launches.py 2>&1 | tee "$DST/asan.log" | tail -3
if grep -q "runtime error\|AddressSanitizer" "$DST/asan.log"; then echo "SANITIZER REPORTS: see $DST/asan.log"; exit 1; fi
echo "no sanitizer report"
