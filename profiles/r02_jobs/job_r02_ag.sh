O=gpurun_out/r02_ag; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" timeout 120 python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
( run "" "5 100000 64 256"; run "" "7 3000 64 256" ) > $O/small.log 2>&1; cat $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for cfg in "5 100000 1024 256" "6 100000 1024 256" "7 100000 1024 256" "5 30000 1024 256" "5 100000 4096 256"; do run "" "$cfg"; done ) > $O/wmbench.log 2>&1
grep -v "^==" $O/wmbench.log
timeout 600 python tools/psetbench.py 1024 2000 5 20 256 2>&1 | grep -v amdgpu
