#!/bin/bash
# tools/fuzz_job.sh TAG -- differential fuzzing of every engine and forced plan on the GPU (tests/fuzz_gpu.py), with the
# automaton kernels' instantiation knobs forced; stops at the first failing run
O=gpurun_out/$1; mkdir -p $O
f() { echo "== AC_TUNE=$1 WM_TUNE=$2 cases=$3 seed=$4 big=$5"; SMH_AC_TUNE="$1" SMH_WM_TUNE="$2" FUZZ_BIG="$5" timeout -k 10 170 python tests/fuzz_gpu.py $3 $4 2>&1 | grep -v amdgpu.ids | tail -2; return ${PIPESTATUS[0]}; }
{ f "" "" 60 31001 && f "clamp=1" "" 45 31002 && f "nch=3" "" 45 31003 && f "nch=2" "" 45 31004 && f "" "pairwg=2" 45 31005 && f "" "" 12 31006 1 && f "clamp=1,nch=2" "" 10 31007 1 && f "" "regv=1" 45 31008 && f "" "regv=0" 25 31009 && f "" "regv=1" 12 31010 1 && f "" "gram=5" 45 31011 && f "" "gram=5,regv=1" 30 31012 && f "" "gram=5" 12 31013 1 && f "" "gram=6" 45 31014 && f "" "gram=6" 12 31015 1; } > $O/fuzz.log 2>&1
rc=$?
grep "^==\|fuzz:\|Error\|assert\|Traceback\|fault" $O/fuzz.log
exit $rc
