#!/bin/bash
# tools/final_job.sh TAG -- the end-of-round evidence in one call: GPU tests, the plain bench, the one-card rehearsals of both
# N > 1 flows, the rocprofv3 passes (kernel trace, SQ counters, FETCH / WRITE sizes), the differential fuzz.
TAG=${1:-final}
O=gpurun_out/$TAG; mkdir -p $O
R=${GRAFT_REPO_ROOT:-/root/repo}
echo "== tests"; timeout -k 10 1100 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
grep -q "failed\|error" $O/pytest_gpu.log && { echo "tests failed"; exit 1; }
echo "== bench"; timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python tools/bench_summary.py $O/bench.json > $O/bench_summary.txt 2>&1; head -3 $O/bench_summary.txt
echo "== rehearse2 (torch ranks on one card)"; timeout -k 10 600 python bench.py --gpus 2 --share-device --mib-per-gpu 256 --shard-mib 512 --verify-mib 160 --steps 3 --warmup 1 --no-skewed > $O/rehearse2_share_device.json 2> $O/rehearse2.err || { tail -5 $O/rehearse2.err; exit 1; }
echo "== native leg, 4 logical shards on one card"; SMH_MULTI_SHARE_DEVICE=1 timeout -k 10 600 python bench.py --multi-leg 4 --steps 3 --mib-per-gpu 256 --shard-mib 512 > $O/smh_multi_share_device_4.json 2> $O/multi4.err || { tail -5 $O/multi4.err; exit 1; }
echo "== counters"; bash tools/collect_counters.sh $TAG > $O/collect.log 2>&1 || { tail -5 $O/collect.log; exit 1; }
head -40 $O/pmc_sq_summary.txt
echo "== fuzz"; timeout -k 10 600 python tests/fuzz_gpu.py 40 4242 > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
echo "job $TAG done"
