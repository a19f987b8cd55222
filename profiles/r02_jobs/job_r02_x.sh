O=gpurun_out/r02_x; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" timeout 120 python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
( run "" "16 1000 64 4"; run "" "12 100000 64 256"; run "" "16 8000 64 4" ) > $O/small.log 2>&1; cat $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for cfg in "12 100000 1024 256" "20 100000 1024 256" "8 100000 1024 256" "16 1000 1024 4"; do run "" "$cfg"; done; run "gram=1" "16 8000 1024 4" ) > $O/wmbench.log 2>&1
grep -v "^==" $O/wmbench.log
bash tools/collect_counters.sh r02_d > $O/collect.log 2>&1
cat gpurun_out/r02_d/pmc_sq_summary.txt | head -120
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r02_d/kernel_stats.csv')))
for r in rows[:24]: print(r['Name'][:120], r['Calls'], r['AverageNs'])
PY
cat gpurun_out/r02_d/hbm_traffic.json | head -80
