O=gpurun_out/r02_ab; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" timeout 120 python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
( for cfg in "16 1000 4096 4" "16 4000 4096 4" "16 8000 4096 4" "16 1000 256 4" "16 4000 256 4" "16 8000 256 4"; do run "gram=1" "$cfg"; done ) > $O/wmbench.log 2>&1
grep -v "^==" $O/wmbench.log
