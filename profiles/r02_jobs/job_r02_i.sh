O=gpurun_out/r02_i; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
for cfg in "16 8000 1024 4" "32 8000 1024 4"; do for t in "gram=1" "gram=3"; do run "$t" "$cfg"; done; done > $O/wmbench.log 2>&1
for cfg in "12 100000 1024 256" "20 100000 1024 256" "8 100000 1024 256"; do for t in "gram=2" "gram=0"; do run "$t" "$cfg"; done; done >> $O/wmbench.log 2>&1
run "" "8 10000 1024 4" >> $O/wmbench.log 2>&1
run "" "12 3000 1024 4" >> $O/wmbench.log 2>&1
cat $O/wmbench.log
if grep -q "Memory access fault" $O/wmbench.log; then echo FAULT; exit 1; fi
