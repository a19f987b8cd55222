O=gpurun_out/r02_am; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" timeout 120 python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
( run "debug" "5 100000 64 256" ) > $O/small.log 2>&1; cat $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
( for cfg in "5 100000 1024 256" "6 100000 1024 256" "7 100000 1024 256" "5 100000 4096 256"; do run "" "$cfg"; done ) > $O/wmbench.log 2>&1
grep -v "^==" $O/wmbench.log
