#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
run() { SMH_AC_TUNE="$1" timeout -k 10 120 python tools/acbench.py $2 $3 $4 2>&1 | grep -v amdgpu.ids | tail -1; }
{ for pf in 0 2; do for m in 8 16 32; do run "pf=$pf" $m 1000 1024; done; run "pf=$pf" 8 8000 4096; run "pf=$pf" 16 8000 1024; done
} > $O/early.log 2>&1; cat $O/early.log
