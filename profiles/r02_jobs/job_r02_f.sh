O=gpurun_out/r02_f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_cfg.log 2>&1; echo "pytest cfg rc $?"; tail -15 $O/pytest_cfg.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_f/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for k in ('ac','wm','ac_8000_patterns','wm_ascii','parity','cpu_baseline','cpu_baseline_wm','cpu_baseline_all_cores','verified','host_pointer_path'):
    print(k, d.get(k))
PY
