O=gpurun_out/r02_at; mkdir -p $O
( for rep in 1 2 3; do for cfg in "32 1000 1024 4" "16 1000 1024 4"; do for t in "lane0=0" "x"; do echo -n "$t "; SMH_WM_TUNE=$t timeout 120 python tools/wmbench.py $cfg 2>&1 | grep -v amdgpu; done; done; done ) > $O/bench.log 2>&1
cat $O/bench.log
