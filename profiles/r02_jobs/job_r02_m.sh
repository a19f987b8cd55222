O=gpurun_out/r02_m; mkdir -p $O
for a in "1024 1000 8 32 4" "1024 300 12 24 4" "256 2000 3 12 20"; do python tools/psetbench.py $a; SMH_PSET_TUNE=classes python tools/psetbench.py $a | grep "pset AC"; done > $O/psetbench.log 2>&1
grep -v amdgpu.ids $O/psetbench.log
if grep -q "Memory access fault" $O/psetbench.log; then echo FAULT; exit 1; fi
