O=gpurun_out/r02_ah; mkdir -p $O
( for cfg in "16 1000 1024" "12 1000 1024 3 $((12 | 9<<8))" "12 1000 1024" "16 1000 1024 3 $((12 | 9<<8))" "16 1000 1024 3 $((12 | 8<<8))" "13 1000 1024 3 $((12 | 9<<8))" "32 1000 1024"; do timeout 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done ) > $O/acbench.log 2>&1
cat $O/acbench.log
