O=gpurun_out/r02_az; mkdir -p $O
( for rep in 1 2; do for cfg in "8 1000 1024" "8 8000 1024" "6 1000 1024" "10 1000 1024" "16 1000 1024 1 8"; do for t in x pf=0; do SMH_AC_TUNE=$t timeout 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done; done; done ) > $O/acbench.log 2>&1
cat $O/acbench.log
