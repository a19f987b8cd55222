O=gpurun_out/r02_z; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" timeout 120 python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
( for p in 1000 2000 3000 4000 6000 8000; do run "gram=1,debug" "16 $p 1024 4"; done
  for t in "gram=1,hd=1,stmin=1" "gram=1,hd=1,stmin=200" "gram=1,hd=0" "gram=1,hd=1,stmin=200,drain=32"; do run "$t" "16 8000 1024 4"; run "$t" "16 4000 1024 4"; done ) > $O/wmbench.log 2>&1
grep -v "^==" $O/wmbench.log
