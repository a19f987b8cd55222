#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
run() { SMH_AC_TUNE="$1" timeout -k 10 120 python tools/acbench.py $2 $3 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -1; }
{ run "" 8 1000 1024; run "" 8 1500 1024; run "" 8 1800 1024; run "" 8 2000 1024
  run "nch=1" 12 1000 1024; run "nch=2" 12 1000 1024; run "" 12 1000 1024
  run "nch=1" 16 1000 1024; run "nch=2" 16 1000 1024
  run "nch=2" 8 1000 1024; run "nch=2" 8 1800 1024
} > $O/matrix.log 2>&1; cat $O/matrix.log
