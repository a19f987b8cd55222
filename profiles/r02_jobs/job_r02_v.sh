O=gpurun_out/r02_v; mkdir -p $O
( timeout 120 python tools/psetbench.py 64 1000 12 32 4; SMH_WM_TUNE=grouped=force timeout 120 python tools/psetbench.py 64 1000 8 32 4 ) > $O/small.log 2>&1; grep -v amdgpu $O/small.log
if grep -q "Memory access fault\|Traceback" $O/small.log; then echo FAULT; exit 1; fi
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
if grep -q "Memory access fault" $O/pytest.log; then echo FAULT; exit 1; fi
( for cfg in "1024 1000 8 32 4" "1024 1000 10 32 4" "1024 1000 12 32 4" "1024 1000 14 32 4" "1024 1000 16 64 4" "1024 200 8 32 4" "1024 4000 12 40 4"; do timeout 200 python tools/psetbench.py $cfg; done
  SMH_WM_TUNE=grouped=force timeout 200 python tools/psetbench.py 1024 1000 8 32 4 ) > $O/psetbench.log 2>&1
grep -v amdgpu $O/psetbench.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02_v/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
for k in ('mixed_8_32','wm_long','wm_ascii'):
    print(k, d.get(k))
print('verified', d['verified']['all_equal'], d['verified']['seconds'])
PY
