O=gpurun_out/r02_ai; mkdir -p $O
( for cfg in "12 1000 1024" "16 1000 1024" "32 1000 1024"; do timeout 120 python tools/wavetrace.py $cfg 2>&1 | grep -v amdgpu | tail -8; done ) > $O/wavetrace.log 2>&1
cat $O/wavetrace.log
