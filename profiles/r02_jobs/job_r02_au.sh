O=gpurun_out/r02_au; mkdir -p $O
( for cfg in "16 1000 1024 4" "32 1000 1024 4" "32 2000 1024 4" "32 3000 1024 4" "32 8000 1024 4"; do timeout 120 python tools/wm_ab.py $cfg lane0=0 lane0=1 2>&1 | grep -v "amdgpu\|in order"; SMH_WM_TUNE=x timeout 100 python tools/wmbench.py $cfg | grep -v amdgpu; done ) > $O/bench.log 2>&1
cat $O/bench.log
