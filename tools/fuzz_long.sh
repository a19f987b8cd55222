#!/bin/bash
# long differential fuzz of the final build: default knobs, fresh seeds, small and big sets
O=gpurun_out/$1; mkdir -p $O
f() { echo "== AC_TUNE=$1 WM_TUNE=$2 cases=$3 seed=$4 big=$5"; SMH_AC_TUNE="$1" SMH_WM_TUNE="$2" FUZZ_BIG="$5" timeout -k 10 280 python tests/fuzz_gpu.py $3 $4 2>&1 | grep -v amdgpu.ids | tail -2; return ${PIPESTATUS[0]}; }
{ f "" "" 120 41001 && f "" "" 120 41002 && f "" "regv=1" 100 41003 && f "" "" 30 41004 1 && f "" "regv=1" 30 41005 1 && f "" "regv=0" 30 41006 1; } > $O/fuzz.log 2>&1
rc=$?
grep "^==\|fuzz:\|Error\|assert\|Traceback\|fault" $O/fuzz.log
exit $rc
