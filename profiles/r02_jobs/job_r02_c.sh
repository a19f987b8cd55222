O=gpurun_out/r02_c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for t in "nch=1" "nch=2"; do
  for cfg in "8 1000 1024" "16 1000 1024" "32 1000 1024" "16 1000 1024 3 2316" "16 1000 4096" "8 1000 4096"; do SMH_AC_TUNE=$t python tools/acbench.py $cfg; done
done > $O/acbench.log 2>&1
python tools/wmbench.py 8 10000 1024 4 >> $O/acbench.log 2>&1
for cfg in "8 1000 1024" "16 1000 1024" "32 1000 1024"; do python tools/wavetrace.py $cfg; done > $O/wavetrace.log 2>&1
SMH_AC_TUNE="nch=2" python tools/wavetrace.py 8 1000 1024 >> $O/wavetrace.log 2>&1
grep -v amdgpu.ids $O/acbench.log; grep -v amdgpu.ids $O/wavetrace.log
