O=gpurun_out/r02_ao; mkdir -p $O
( for cfg in "16 1000 1024" "16 1000 1024 3 $((12 | 9<<8))" "16 1000 1024 3 $((12 | 8<<8))" "16 1000 1024 3 $((12 | 7<<8))" "16 1000 1024 3 $((16 | 9<<8))" "16 1000 1024 3 $((16 | 8<<8))"; do timeout -k 5 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu | tail -1; done ) > $O/acbench.log 2>&1
cat $O/acbench.log
