#!/bin/bash
# round-3 experiment: the hybrid automaton kernel with 8 / 12 waves per CU and 3..6 chains per lane (build with -DSMH_AC_EXPERIMENT)
O=gpurun_out/$1; mkdir -p $O
run() { SMH_AC_TUNE="$1" timeout -k 10 120 python tools/acbench.py $2 1000 1024 2>&1 | grep -v amdgpu.ids | tail -1; }
{
for m in 16 32; do
  run "" $m
  run "nch=6,bt=512,pf=0" $m
  run "nch=6,bt=512,pf=1" $m
  run "nch=5,bt=512,pf=1" $m
  run "nch=4,bt=512,pf=1" $m
  run "nch=4,bt=512,pf=0" $m
  run "nch=3,bt=512,pf=1" $m
  run "nch=4,bt=768,pf=0" $m
  run "nch=4,bt=768,pf=1" $m
done
} > $O/fatwaves.log 2>&1
cat $O/fatwaves.log
