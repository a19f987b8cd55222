O=gpurun_out/r02_t; mkdir -p $O
run() { echo "== SMH_WM_TUNE=$1 :: $2"; SMH_WM_TUNE="$1" timeout 120 python tools/wmbench.py $2 2>&1 | grep -v amdgpu.ids; }
( run "" "5 100000 64 256"; run "" "20 1000 64 256" ) > $O/small.log 2>&1; cat $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for st in "" "stage=0"; do for cfg in "5 100000 1024 256" "6 100000 1024 256" "7 100000 1024 256" "5 10000 1024 256"; do run "$st" "$cfg"; done; done ) > $O/wmbench.log 2>&1
cat $O/wmbench.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_t/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for k in ('ac','wm','wm_long','ac_8000_patterns','wm_ascii','stream_read','positions','cpu_baseline','cpu_baseline_wm'):
    print(k, d.get(k))
print('verified', d['verified']['all_equal'], d['verified']['seconds'])
PY
