O=gpurun_out/r02_aq; mkdir -p $O
( for cfg in "32 1000 1024 4" "32 1000 4096 4" "32 8000 1024 4" "32 8000 4096 4" "16 8000 1024 4" "16 8000 4096 4"; do timeout 120 python tools/wmbench.py $cfg 2>&1 | grep -v amdgpu; done
  for cfg in "8 1000 1024" "8 1000 4096" "16 1000 4096"; do timeout 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done ) > $O/bench.log 2>&1
cat $O/bench.log
