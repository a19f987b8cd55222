O=gpurun_out/r02_ar; mkdir -p $O
( for t in "debug" "stage=0" "hd=1" "hd=1,stmin=1"; do for cfg in "32 8000 1024 4" "16 8000 1024 4" "32 4000 1024 4" "32 2000 1024 4"; do echo "tune=$t"; SMH_WM_TUNE=$t timeout 120 python tools/wmbench.py $cfg 2>&1 | grep -v amdgpu; done; done ) > $O/bench.log 2>&1
cat $O/bench.log
