#!/bin/bash
# Experiment (round 3): in-register verify of the pair-gram kernels against the staged verify, same handle, launches interleaved.
# usage (GPU box): bash tools/exp_regv.sh TAG [sets...]   (a set = "m p MiB alphabet")
TAG=$1; shift; O=gpurun_out/$TAG; mkdir -p $O
[ $# -eq 0 ] && set -- "16 8000 1024 4" "32 8000 1024 4" "16 1000 1024 4" "32 1000 1024 4" "24 3000 1024 4" "16 20000 1024 4" "12 4000 1024 4"
for s in "$@"; do
  SMH_WM_TUNE=debug timeout -k 10 300 python tools/wm_ab.py $s regv=0 regv=1 2>&1 | grep -v "in order\|amdgpu.ids" | tee -a $O/regv.log || exit 1
done
