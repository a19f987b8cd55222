set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_d
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in cuckoo bucket bucket_noover; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc1_$v -- python3 $R/tools/key_pmc.py $v > $OUT/pmc1_$v.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc2_$v -- python3 $R/tools/key_pmc.py $v > $OUT/pmc2_$v.log 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_VALU --output-format csv -d $OUT/pmc3_$v -- python3 $R/tools/key_pmc.py $v > $OUT/pmc3_$v.log 2>&1 || true
  echo "== $v" >> $OUT/summary.txt
  python3 $R/tools/pmc_summary.py $OUT/pmc1_$v $OUT/pmc2_$v $OUT/pmc3_$v >> $OUT/summary.txt 2>&1
done
find $OUT -name "*counter_collection.csv" -size +2M -delete
cat $OUT/summary.txt
