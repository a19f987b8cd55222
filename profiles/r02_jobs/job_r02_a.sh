mkdir -p gpurun_out/r02_a
python -m pytest tests -m gpu -x -q > gpurun_out/r02_a/pytest.log 2>&1; echo "pytest rc $?"
tools/clockprobe > gpurun_out/r02_a/clock.log 2>&1
bash tools/collect_counters.sh r02_a > gpurun_out/r02_a/collect.log 2>&1
for cfg in "16 1000 1024" "16 1000 1024 1 16" "16 1000 1024 3 2316" "16 1000 1024 2 8" "32 1000 1024" "32 1000 1024 2 8" "32 1000 1024 1 24" "8 1000 1024" "16 1000 4096" "8 1000 4096"; do python tools/acbench.py $cfg; done > gpurun_out/r02_a/acbench.log 2>&1
SMH_AC_TUNE="bpc=2" python tools/acbench.py 8 1000 1024 >> gpurun_out/r02_a/acbench.log 2>&1
tail -3 gpurun_out/r02_a/pytest.log; cat gpurun_out/r02_a/clock.log gpurun_out/r02_a/acbench.log
