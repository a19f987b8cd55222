#!/bin/bash
# tools/pmc_ac.sh TAG "TUNE" M P MIB [STRIDE DEPTH] -- two SQ counter passes over tools/acbench.py for one kernel configuration
TAG=$1; TUNE=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
NAME=$(echo "pmc_${TUNE}_$*" | tr ' =,' '___')
cd /tmp && export TMPDIR=/tmp && export SMH_AC_TUNE="$TUNE"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/${NAME}_1 -- python3 $R/tools/acbench.py "$@" > $O/${NAME}_1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/${NAME}_2 -- python3 $R/tools/acbench.py "$@" > $O/${NAME}_2.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/${NAME}_3 -- python3 $R/tools/acbench.py "$@" > $O/${NAME}_3.log 2>&1
echo "== $NAME"; python3 $R/tools/pmc_summary.py $O/${NAME}_1 $O/${NAME}_2 $O/${NAME}_3 | grep -v "^    .*n=1 " 
find $O -name "*counter_collection.csv" -size +4M -delete
