O=gpurun_out/r02_ao; mkdir -p $O
( for rep in 1 2; do for t in x nch=4; do for cfg in "16 1000 1024" "32 1000 1024"; do SMH_AC_TUNE=$t timeout -k 5 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done; done; done ) > $O/acbench.log 2>&1
cat $O/acbench.log
