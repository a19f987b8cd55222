O=gpurun_out/r02_al; mkdir -p $O
( timeout 120 python tools/acbench.py 16 1000 64; timeout 120 python tools/acbench.py 16 1000 64 1 8 ) > $O/small.log 2>&1; grep -v amdgpu $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for cfg in "16 1000 1024" "32 1000 1024" "16 1000 1024 1 8" "16 1000 1024 1 12" "16 1000 1024 2 8" "8 1000 1024"; do timeout 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done ) > $O/acbench.log 2>&1
cat $O/acbench.log
f() { echo "== $1 :: $2 $3"; SMH_WM_TUNE="$1" FUZZ_BIG="$4" timeout -k 10 170 python tests/fuzz_gpu.py $2 $3 2>&1 | grep -v amdgpu.ids | tail -1; }
( f "" 60 9001; f "" 60 9002 ) 2>&1
