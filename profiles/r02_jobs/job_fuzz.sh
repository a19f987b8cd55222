O=gpurun_out/fuzz; mkdir -p $O
f() { echo "== $1 :: $2 $3"; SMH_WM_TUNE="$1" FUZZ_BIG="$4" timeout -k 10 170 python tests/fuzz_gpu.py $2 $3 2>&1 | grep -v amdgpu.ids | tail -3; }
( f "" 60 7001; f "" 60 7002; f "gram=1,hd=1,stmin=1" 40 7003; f "gram=1,hd=0" 40 7004; f "gram=2,hd=1,stmin=2" 40 7005; f "gram=2,hd=0" 40 7006; f "gram=3,hd=0" 30 7007; f "" 12 7008 1; f "grouped=force" 45 7009; f "gram=1,hd=1,stmin=200" 30 7010 ) > $O/fuzz.log 2>&1
cat $O/fuzz.log
if grep -q "Memory access fault\|Traceback\|Error" $O/fuzz.log; then echo FUZZ-FAIL; fi
