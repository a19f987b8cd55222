O=gpurun_out/r02_ak; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
f() { echo "== $1 :: $2 $3"; SMH_WM_TUNE="$1" FUZZ_BIG="$4" timeout -k 10 170 python tests/fuzz_gpu.py $2 $3 2>&1 | grep -v amdgpu.ids | tail -2; }
( f "" 60 8001; f "" 60 8002; f "" 12 8003 1 ) > $O/fuzz.log 2>&1
cat $O/fuzz.log
timeout 300 python tools/leakcheck.py 2>&1 | grep -v amdgpu | tail -3
