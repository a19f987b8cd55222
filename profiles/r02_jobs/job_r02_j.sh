O=gpurun_out/r02_j; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
for cfg in "16 8000 1024 4"; do for t in "gram=1,drain=64" "gram=1,drain=16" "gram=1,drain=1"; do run "$t" "$cfg"; done; done > $O/wmbench.log 2>&1
for cfg in "12 100000 1024 256" "20 100000 1024 256"; do for t in "gram=2,drain=64" "gram=2,drain=24" "gram=2,drain=1"; do run "$t" "$cfg"; done; done >> $O/wmbench.log 2>&1
cat $O/wmbench.log
if grep -q "Memory access fault" $O/wmbench.log; then echo FAULT; exit 1; fi
