O=gpurun_out/r02_u; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" timeout 120 python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
( run "" "5 100000 64 256"; run "" "12 100000 64 256"; run "" "16 8000 64 4" ) > $O/small.log 2>&1; cat $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for st in 1 4 8 16 32 200; do
    for cfg in "12 100000 1024 256" "20 100000 1024 256" "5 100000 1024 256" "5 10000 1024 256" "12 10000 1024 256"; do run "stmin=$st" "$cfg"; done
    for cfg in "16 8000 1024 4" "16 3000 1024 4"; do run "gram=1,stmin=$st" "$cfg"; done
  done ) > $O/wmbench.log 2>&1
cat $O/wmbench.log
