# PMC counters of the Voicebox NFE kernels (separate passes, no tracing)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
: > $R/gpurun_out/r03_vb_pmc2.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES"; do
  rm -rf $R/gpurun_out/pmc_vb
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_vb -- python3 $R/tools/vb_nfe.py 3 > /dev/null 2>&1
  f=$(ls $R/gpurun_out/pmc_vb/*/*counter_collection.csv 2>/dev/null | head -1)
  echo "== $set" >> $R/gpurun_out/r03_vb_pmc2.txt
  for k in "gemm_kernel<unsigned short, 256, 128" "gemm_kernel<unsigned short, 288, 128" "gemm_kernel<unsigned short, 128, 128" "attn_kernel<64" "norm_kernel<4>"; do
    echo "-- $k" >> $R/gpurun_out/r03_vb_pmc2.txt
    python3 $R/tools/pmc_summary.py "$f" "$k" >> $R/gpurun_out/r03_vb_pmc2.txt 2>&1
  done
done
rm -rf $R/gpurun_out/pmc_vb
cat $R/gpurun_out/r03_vb_pmc2.txt
