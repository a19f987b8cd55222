O=gpurun_out/r02_ao; mkdir -p $O
( for rep in 1 2; do for cfg in "16 1000 1024" "32 1000 1024"; do timeout -k 5 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done; done ) > $O/acbench.log 2>&1
cat $O/acbench.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "ac or hybrid or plan or positions" 2>&1 | tail -3
