O=gpurun_out/r02_w; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" timeout 120 python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
( run "gram=1,hd=0" "16 1000 64 4"; run "gram=2,hd=0" "12 100000 64 256"; run "gram=3,hd=0" "16 8000 64 4" ) > $O/small.log 2>&1; cat $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for hd in "hd=0" "hd=1" ""; do
    for cfg in "16 1000 1024 4" "32 1000 1024 4" "16 3000 1024 4" "16 8000 1024 4"; do run "gram=1,$hd" "$cfg"; done
    for cfg in "16 8000 1024 4"; do run "gram=3,$hd" "$cfg"; done
    for cfg in "12 100000 1024 256" "20 100000 1024 256" "12 10000 1024 256" "5 10000 1024 256" "12 1000 1024 256"; do run "gram=2,$hd" "$cfg"; done
  done ) > $O/wmbench.log 2>&1
grep -v "^==" $O/wmbench.log
