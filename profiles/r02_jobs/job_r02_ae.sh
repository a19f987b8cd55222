O=gpurun_out/r02_ae; mkdir -p $O
( timeout 120 python tools/psetbench.py 64 1000 8 32 4 ) > $O/small.log 2>&1; grep -v amdgpu $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 600 python -m pytest tests/test_pattern_sets.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for cfg in "1024 1000 8 32 4" "1024 200 8 32 4" "1024 3000 8 40 4" "1024 300 3 12 20" "1024 500 6 24 8"; do timeout 200 python tools/psetbench.py $cfg; done ) > $O/psetbench.log 2>&1
grep -v amdgpu $O/psetbench.log
