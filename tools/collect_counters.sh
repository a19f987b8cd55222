#!/bin/bash
# tools/collect_counters.sh TAG -- on the GPU box: rocprofv3 passes over `bench.py` for profiles/TAG/.
#   kernel trace + stats, two SQ counter passes, FETCH_SIZE and WRITE_SIZE passes (separate, as
#   MI355X_MICROARCH.md prescribes).  Output lands in gpurun_out/TAG/; copy the summaries to profiles/TAG/.
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-multi --no-skewed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --no-multi --no-skewed > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/pmc_sq1 $OUT/pmc_sq2 > $OUT/pmc_sq_summary.txt 2>&1
python3 $R/tools/make_traffic_json.py $OUT/pmc_fetch $OUT/pmc_write profiles/$TAG > $OUT/hbm_traffic.json 2> $OUT/traffic.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 $R/tools/durations_by_text_size.py $OUT/trace > $OUT/kernel_durations_by_text_size.txt 2>&1
# keep the merge-back small: the raw per-dispatch CSVs are not needed once summarised
find $OUT -name "*counter_collection.csv" -size +8M -delete
find $OUT -name "*kernel_trace.csv" -size +8M -delete
ls -la $OUT
