O=gpurun_out/fuzz3; mkdir -p $O
f() { echo "== $1 :: $2 $3 big=$4"; SMH_WM_TUNE="$1" FUZZ_BIG="$4" timeout -k 10 175 python tests/fuzz_gpu.py $2 $3 2>&1 | grep -v amdgpu.ids | tail -2; }
( f "" 70 13001; f "" 70 13002; f "" 14 13003 1; f "gram=1,lane0=1" 50 13004; f "gram=1,lane0=1,hd=1,stmin=1" 40 13005; f "gram=1,lane0=0" 40 13006; f "gram=1,lane0=1,stage=0" 30 13007; f "lane0=1" 14 13008 1; f "grouped=force" 30 13009 ) > $O/fuzz.log 2>&1
grep "^==\|fuzz:\|Error\|assert\|Traceback\|fault" $O/fuzz.log
