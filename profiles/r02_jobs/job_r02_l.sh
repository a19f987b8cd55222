O=gpurun_out/r02_l; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
python tools/psetbench.py > $O/psetbench.log 2>&1; SMH_PSET_TUNE=classes python tools/psetbench.py >> $O/psetbench.log 2>&1
grep -v amdgpu.ids $O/psetbench.log
