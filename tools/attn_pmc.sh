# PMC counters of the two Voicebox attention kernels (separate passes, no tracing): AB_VARIANTS picks the kernel
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/attn_pmc.txt
: > $OUT
for v in 0 1; do
export AB_VARIANTS=$v
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT"; do
  rm -rf $R/gpurun_out/pmc_at
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_at -- python3 $R/tools/attn_bench.py > /dev/null 2>&1
  f=$(ls $R/gpurun_out/pmc_at/*/*counter_collection.csv 2>/dev/null | head -1)
  echo "== V16=$v $set" >> $OUT
  python3 $R/tools/pmc_summary.py "$f" "attn" >> $OUT 2>&1
done
done
rm -rf $R/gpurun_out/pmc_at
cat $OUT
