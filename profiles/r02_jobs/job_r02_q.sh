O=gpurun_out/r02_q; mkdir -p $O
( timeout 120 python tools/acbench.py 8 1000 1 ; timeout 120 python tools/acbench.py 16 1000 64 ) > $O/small.log 2>&1; echo "small rc $?"; grep -v amdgpu $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 300 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
for t in "gsched=1" "gsched=0"; do
  for cfg in "8 1000 1024" "16 1000 1024" "32 1000 1024" "8 1000 4096"; do SMH_AC_TUNE=$t timeout 120 python tools/acbench.py $cfg; done
done > $O/acbench.log 2>&1
grep -v amdgpu.ids $O/acbench.log
timeout 120 python tools/wavetrace.py 16 1000 1024 | tail -7
