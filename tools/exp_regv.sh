#!/bin/bash
# Experiment (round 3): in-register verify of the pair-gram kernels against the staged verify, same handle, launches interleaved.
# usage (GPU box): bash tools/exp_regv.sh TAG
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
for s in "16 8000 1024 4" "32 8000 1024 4" "16 1000 1024 4" "32 1000 1024 4" "24 3000 1024 4" "16 20000 1024 4" "12 4000 1024 4"; do
  timeout -k 10 300 python tools/wm_ab.py $s regv=0 regv=1 2>&1 | grep -v "in order" | tee -a $O/regv.log || exit 1
done
