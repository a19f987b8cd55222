O=gpurun_out/r02_af; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for cfg in "1024 1000 8 32 4" "1024 200 8 32 4" "1024 3000 8 40 4" "1024 8000 8 32 4" "1024 500 6 24 8" "1024 1000 12 32 4" "1024 2000 5 20 256"; do timeout 300 python tools/psetbench.py $cfg; done ) > $O/psetbench.log 2>&1
grep -v amdgpu $O/psetbench.log
