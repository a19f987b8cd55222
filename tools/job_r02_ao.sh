O=gpurun_out/r02_ao; mkdir -p $O
( for t in "" "nch=2"; do for cfg in "16 1000 1024" "32 1000 1024"; do SMH_AC_TUNE=$t timeout 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done; done
  for t in "" "nch=2" "nch=3"; do SMH_AC_TUNE=$t timeout 120 python tools/acbench.py 8 1000 1024 2>&1 | grep -v amdgpu; done ) > $O/acbench.log 2>&1
cat $O/acbench.log
