O=gpurun_out/r02_b; mkdir -p $O
for t in "nch=1" "nch=2"; do
  for cfg in "8 1000 1024" "16 1000 1024" "32 1000 1024" "16 1000 1024 3 2316" "16 1000 4096" "8 1000 4096"; do SMH_AC_TUNE=$t python tools/acbench.py $cfg; done
done > $O/acbench.log 2>&1
for cfg in "8 1000 1024" "16 1000 1024" "32 1000 1024"; do python tools/wavetrace.py $cfg; done > $O/wavetrace.log 2>&1
SMH_AC_TUNE="nch=2" python tools/wavetrace.py 8 1000 1024 >> $O/wavetrace.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_BUSY_CU_CYCLES --output-format csv -d $R/$O/pmc_lvl -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-wm > $R/$O/pmc_lvl.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/pmc_lvl > $O/pmc_lvl_summary.txt 2>&1
grep -v amdgpu.ids $O/acbench.log; grep -v amdgpu.ids $O/wavetrace.log; cat $O/pmc_lvl_summary.txt
