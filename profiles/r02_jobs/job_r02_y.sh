O=gpurun_out/r02_y; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" timeout 120 python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
( run "" "16 1000 64 4"; run "" "32 8000 64 4"; run "" "16 8000 64 4" ) > $O/small.log 2>&1; cat $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for cfg in "16 1000 1024 4" "32 1000 1024 4" "16 8000 1024 4" "32 8000 1024 4" "24 8000 1024 4" "12 8000 1024 4" "16 30000 1024 4" "32 30000 1024 4"; do run "" "$cfg"; done
  for cfg in "16 8000 1024 4" "16 30000 1024 4" "32 30000 1024 4"; do run "gram=1,hd=0" "$cfg"; run "gram=1,hd=1" "$cfg"; run "gram=3" "$cfg"; done ) > $O/wmbench.log 2>&1
grep -v "^==" $O/wmbench.log
