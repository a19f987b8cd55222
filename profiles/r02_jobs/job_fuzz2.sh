O=gpurun_out/fuzz2; mkdir -p $O
f() { echo "== $1 :: $2 $3 big=$4"; SMH_WM_TUNE="$1" FUZZ_BIG="$4" timeout -k 10 175 python tests/fuzz_gpu.py $2 $3 2>&1 | grep -v amdgpu.ids | tail -2; }
( f "" 70 11001; f "" 70 11002; f "" 70 11003; f "" 14 11004 1; f "" 14 11005 1; f "gram=1,hd=1,stmin=1" 40 11006; f "gram=2,hd=1,stmin=3" 40 11007; f "stage=0" 40 11008; f "grouped=force" 40 11009; f "gram=3,hd=1" 30 11010 ) > $O/fuzz.log 2>&1
grep "^==\|fuzz:\|Error\|assert\|Traceback" $O/fuzz.log
