#!/bin/bash
# tools/gpu_job.sh TAG STEP [STEP...] -- the one job script run on the GPU box (gpurun -- 'bash tools/gpu_job.sh r03_a tests bench').
# Output lands in gpurun_out/TAG/ (merged back by gpurun); copy what is to be judged into profiles/TAG/.
# Steps (a failing step ends the job: no GPU step is started after one that timed out or faulted):
#   tests            pytest -m gpu
#   tests:EXPR       pytest -m gpu -k EXPR
#   bench            python bench.py (defaults: the driver's N=1 run)
#   bench-fast       bench.py without the CPU legs and the one-process leg (kernel numbers only)
#   rehearse2        the N=2 control flow of bench.py on ONE card (--share-device, gloo, small shards)
#   counters         tools/collect_counters.sh TAG (rocprofv3 kernel trace + SQ / FETCH / WRITE passes over bench.py)
#   trace            rocprofv3 --kernel-trace --stats over bench-fast only
#   env:K=V          export K=V for the following steps (SMH_AC_TUNE=..., SMH_WM_TUNE=...)
#   py:FILE[:ARGS]   python FILE ARGS (comma-separated) with output to gpurun_out/TAG/<file>.log
#   sh:FILE          bash FILE TAG (an experiment script that may rebuild the library: put it last)
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
fail() { echo "STEP FAILED: $1"; exit 1; }
for step in "$@"; do
  echo "== $step"
  case $step in
    tests)      timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || fail "pytest rc $rc";;
    tests:*)    timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "${step#tests:}" > $O/pytest_k.log 2>&1; rc=$?; tail -3 $O/pytest_k.log; [ $rc -eq 0 ] || fail "pytest rc $rc";;
    bench)      timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -3 $O/bench.err; python tools/bench_summary.py $O/bench.json; [ $rc -eq 0 ] || fail "bench rc $rc";;
    bench-fast) timeout -k 10 600 python bench.py --no-cpu --no-multi > $O/bench_fast.json 2> $O/bench_fast.err; rc=$?; tail -3 $O/bench_fast.err; python tools/bench_summary.py $O/bench_fast.json; [ $rc -eq 0 ] || fail "bench rc $rc";;
    rehearse2)  timeout -k 10 600 python bench.py --gpus 2 --share-device --mib-per-gpu 256 --shard-mib 512 --verify-mib 160 --steps 3 --warmup 1 > $O/rehearse2.json 2> $O/rehearse2.err; rc=$?; tail -3 $O/rehearse2.err; python tools/bench_summary.py $O/rehearse2.json; [ $rc -eq 0 ] || fail "rehearse2 rc $rc";;
    counters)   bash tools/collect_counters.sh $TAG > $O/collect.log 2>&1 || fail counters; cat $O/pmc_sq_summary.txt | head -60;;
    trace)      ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace -- python3 $R/bench.py --no-cpu --no-multi > $R/$O/trace.log 2>&1 ) || fail trace
                find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; find $O -name "*kernel_trace.csv" -size +8M -delete
                python - <<PY
import csv
for r in list(csv.DictReader(open('$O/kernel_stats.csv')))[:30]: print(r['Name'][:110], r['Calls'], r['AverageNs'])
PY
                ;;
    env:*)      export "${step#env:}";;
    py:*)       IFS=: read -r _ file pargs <<< "$step"; timeout -k 10 900 python $file ${pargs//,/ } > $O/$(basename $file .py).log 2>&1; rc=$?; tail -40 $O/$(basename $file .py).log; [ $rc -eq 0 ] || fail "$file rc $rc";;
    sh:*)       timeout -k 10 1000 bash "${step#sh:}" $TAG; rc=$?; [ $rc -eq 0 ] || fail "${step#sh:} rc $rc";;
    *)          fail "unknown step $step";;
  esac
  if grep -qs "Memory access fault" $O/*.log $O/*.err; then fail "GPU memory access fault"; fi
done
echo "job $TAG done"
