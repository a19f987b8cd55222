#!/bin/bash
# tools/final_job.sh TAG -- the end-of-round evidence in one gpurun call (rounds 5, 6): GPU tests (parity run, then the perf
# expectations enforced), the plain bench, the rocprofv3 passes over the bench INCLUDING the non-uniform corpora (that is where the
# key and window-hash kernels run): kernel trace, two SQ passes, FETCH_SIZE and WRITE_SIZE passes (separate, as
# MI355X_MICROARCH.md prescribes), the differential fuzz.
TAG=${1:-r06_final}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
echo "== tests"; timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
grep -q " failed\| error" $O/pytest_gpu.log && { echo "tests failed"; exit 1; }
echo "== perf expectations enforced"; timeout -k 10 600 python -m pytest tests -m "gpu and perf" -q > $O/pytest_gpu_perf.log 2>&1; tail -2 $O/pytest_gpu_perf.log
echo "== bench"; timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cp bench_detail.json $O/bench_detail.json
echo "== counters"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-multi --no-small"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --no-multi --no-small > $O/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -- $BENCH > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_sq2 -- $BENCH > $O/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $BENCH > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $BENCH > $O/pmc_write.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 > $O/pmc_sq_summary.txt 2>&1
python3 tools/make_traffic_json.py $O/pmc_fetch $O/pmc_write profiles/$TAG > $O/hbm_traffic.json 2> $O/traffic.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python3 tools/durations_by_text_size.py $O/trace > $O/kernel_durations_by_text_size.txt 2>&1
find $O -name "*counter_collection.csv" -size +8M -delete
find $O -name "*kernel_trace.csv" -size +8M -delete
echo "== fuzz"; timeout -k 10 600 python tests/fuzz_gpu.py 40 4242 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
ls $O
echo "job $TAG done"
