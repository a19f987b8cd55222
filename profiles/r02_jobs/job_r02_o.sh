bash tools/collect_counters.sh r02_c > gpurun_out/collect_r02_c.log 2>&1
cat gpurun_out/r02_c/pmc_sq_summary.txt | head -150
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r02_c/kernel_stats.csv')))
for r in rows[:16]: print(r['Name'][:110], r['Calls'], r['AverageNs'])
PY
