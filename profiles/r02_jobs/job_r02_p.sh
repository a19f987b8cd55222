O=gpurun_out/r02_p; mkdir -p $O
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for k in ('ac','wm','ac_8000_patterns','wm_ascii','stream_read','positions','cpu_baseline','cpu_baseline_wm','cpu_baseline_all_cores','host_pointer_path'):
    print(k, d.get(k))
print('verified', d['verified']['all_equal'], d['verified']['seconds'])
PY
for cfg in "16 1000 1024" ; do python tools/wavetrace.py $cfg | tail -7; done
