O=gpurun_out/r02_g; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
for cfg in "16 8000 1024 4" "32 8000 1024 4" "16 1000 1024 4" "12 100000 1024 256" "20 100000 1024 256" "5 100000 1024 256" "8 10000 1024 4" "12 1000 1024 4"; do python tools/wmbench.py $cfg; done > $O/wmbench.log 2>&1
grep -v amdgpu.ids $O/wmbench.log
