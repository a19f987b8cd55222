O=gpurun_out/r02_h; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
for cfg in "16 8000 1024 4" "32 8000 1024 4"; do
  for t in "gram=1" "gram=1,noverify" "gram=3" "gram=3,noverify" "gram=0" "gram=0,noverify"; do run "$t" "$cfg"; done
done > $O/wmbench.log 2>&1
for cfg in "12 100000 1024 256" "20 100000 1024 256" "5 100000 1024 256"; do
  for t in "gram=2" "gram=2,noverify" "gram=0" "gram=0,noverify"; do run "$t" "$cfg"; done
done >> $O/wmbench.log 2>&1
for cfg in "16 1000 1024 4" "12 3000 1024 4"; do for t in "gram=1" "gram=3" "gram=1,noverify"; do run "$t" "$cfg"; done; done >> $O/wmbench.log 2>&1
cat $O/wmbench.log
