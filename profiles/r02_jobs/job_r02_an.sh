O=gpurun_out/r02_an; mkdir -p $O
( for cfg in "16 1000 1024" "16 1000 1024 3 $((11 | 9<<8))" "16 1000 1024 3 $((10 | 9<<8))" "16 1000 1024 3 $((13 | 9<<8))" "32 1000 1024 3 $((11 | 9<<8))" "32 1000 1024"; do timeout 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done ) > $O/acbench.log 2>&1
cat $O/acbench.log
