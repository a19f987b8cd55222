O=gpurun_out/r02_as; mkdir -p $O
( for cfg in "32 4000 1024 4 0 0" "32 4000 1024 4 0 2" "32 4000 1024 4 0 1" "32 8000 1024 4 0 0" "32 3000 1024 4 0 2" "32 6000 1024 4 0 2" "32 6000 1024 4 0 0" "16 4000 1024 4 0 0" "16 8000 1024 4 0 0"; do timeout 120 python tools/wmbench.py $cfg 2>&1 | grep -v amdgpu; done ) > $O/bench.log 2>&1
cat $O/bench.log
