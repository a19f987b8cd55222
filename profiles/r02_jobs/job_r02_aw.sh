O=gpurun_out/r02_aw; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
timeout 600 python bench.py --no-cpu > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_aw/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac']); print(d['positions']); print(d['stream_read'])
PY
