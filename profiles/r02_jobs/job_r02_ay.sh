O=gpurun_out/r02_ay; mkdir -p $O
for rep in 1 2; do for t in x nch=3 nch=4; do SMH_AC_TUNE=$t timeout 300 python bench.py --no-cpu --steps 30 > $O/b_${t}_$rep.json 2>/dev/null; done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r02_ay/b_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['ac']['m8']['kernel_ms'], d['ac']['m16']['kernel_ms'], d['ac']['m32']['kernel_ms'])
PY
