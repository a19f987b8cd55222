#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
run() { SMH_AC_TUNE="$1" timeout -k 10 120 python tools/acbench.py $2 $3 $4 $5 $6 2>&1 | grep -v amdgpu.ids | tail -1; }
{ for t in "pk=0" "pk=1,pkpf=1" "pk=2,pkpf=1" "pk=3,pkpf=1" "pk=4,pkpf=1" "pk=2,pkpf=0" "pk=3,pkpf=0" "pk=4,pkpf=0"; do for m in 8 16 32; do run "$t" $m 1000 1024; done; done
} > $O/packed.log 2>&1; cat $O/packed.log
