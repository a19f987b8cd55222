O=gpurun_out/r02_aj; mkdir -p $O
( timeout 120 python tools/acbench.py 16 1000 64; timeout 120 python tools/acbench.py 32 1000 64 ) > $O/small.log 2>&1; grep -v amdgpu $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for cfg in "16 1000 1024" "32 1000 1024" "12 1000 1024" "13 1000 1024 3 $((12 | 9<<8))" "24 3000 1024" "16 1000 4096"; do timeout 120 python tools/acbench.py $cfg 2>&1 | grep -v amdgpu; done
  for cfg in "16 1000 1024" "32 1000 1024"; do timeout 120 python tools/wavetrace.py $cfg 2>&1 | grep -v amdgpu | tail -7 | head -4; done ) > $O/acbench.log 2>&1
cat $O/acbench.log
