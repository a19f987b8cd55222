O=gpurun_out/r02_bb; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
bash tools/collect_counters.sh r02_h > $O/collect.log 2>&1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
cp $O/bench.json gpurun_out/r02_h/bench.json
python - <<'PY'
import json, csv
d=json.loads(open('gpurun_out/r02_bb/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for k in ('ac','wm','wm_long','mixed_8_32','ac_8000_patterns','wm_ascii','stream_read','positions','cpu_baseline','cpu_baseline_wm','cpu_baseline_all_cores','host_pointer_path'):
    print(k, d.get(k))
print('verified', d['verified']['all_equal'], d['verified']['seconds'])
rows=list(csv.DictReader(open('gpurun_out/r02_h/kernel_stats.csv')))
for r in rows[:24]: print(r['Name'][:100], r['Calls'], r['AverageNs'])
PY
