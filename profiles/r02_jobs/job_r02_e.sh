O=gpurun_out/r02_e; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for cfg in "16 1000 1024" "32 1000 1024" "16 1000 1024 3 2060" "16 1000 1024 3 2570" "24 1000 1024" "12 1000 1024" "16 200 1024" "16 3000 1024"; do python tools/acbench.py $cfg; done > $O/acbench.log 2>&1
python tools/wavetrace.py 16 1000 1024 | tail -7 > $O/wavetrace.log 2>&1
grep -v amdgpu.ids $O/acbench.log; cat $O/wavetrace.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_e/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for k in ('ac','wm','ac_8000_patterns','wm_ascii','parity','stream_read','positions'):
    print(k, d.get(k))
PY
