#!/bin/bash
# Experiment (round 3): register prefetch of the next chunk in the WM kernels on the survivor-heavy sets, where a wave
# spends a few LDS round trips per chunk in the staged verify with nothing of its own in flight.
# usage (GPU box): bash tools/exp_prefetch.sh TAG
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
P=cuda-aho-corasick-wu-manber_amd
run() {
  for s in "16 8000 1024 4" "32 8000 1024 4" "16 1000 1024 4" "12 100000 1024 256" "20 100000 1024 256" "8 100000 1024 256" "5 100000 1024 256"; do
    timeout -k 10 300 python tools/wmbench.py $s || exit 1
  done
}
echo "== default build (SMH_PREFETCH=0)" | tee $O/prefetch.log
run >> $O/prefetch.log 2>&1 || { tail -5 $O/prefetch.log; exit 1; }
echo "== wm kernels rebuilt with -DSMH_PREFETCH=1" | tee -a $O/prefetch.log
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -std=c++17 -Wall -Wno-unused-function -DSMH_PREFETCH=1 -c $P/csrc/wm_kernels.hip -o $P/build/wm_kernels.o >> $O/prefetch.log 2>&1 || exit 1
touch $P/build/wm_kernels.o
( cd $P && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o libsmatcher_hip.so build/*.o -lm -ldl ) >> $O/prefetch.log 2>&1 || exit 1
run >> $O/prefetch.log 2>&1 || { tail -5 $O/prefetch.log; exit 1; }
grep "^==\|^WM" $O/prefetch.log
