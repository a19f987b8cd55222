#!/bin/bash
O=gpurun_out/$1; mkdir -p $O
{ for m in 8 16 32; do timeout -k 10 120 python tools/acbench.py $m 1000 1024 2>&1 | grep -v amdgpu.ids | tail -1; done
  timeout -k 10 120 python tools/acbench.py 8 8000 4096 2>&1 | grep -v amdgpu.ids | tail -1
} > $O/l2touch_$2.log 2>&1; cat $O/l2touch_$2.log
