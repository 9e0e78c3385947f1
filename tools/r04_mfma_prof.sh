# per-kernel durations + HBM bytes of the matrix-core batched decode (B = 16): bash tools/r04_mfma_prof.sh
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mfma -- python3 $R/tools/batch_rate.py 16 32 > $R/gpurun_out/r04_mfma_prof.out 2>&1
f=$(find $R/gpurun_out/prof_mfma -name "*kernel_stats.csv" | head -1)
python3 $R/tools/prof_summary.py $f 25 > $R/gpurun_out/r04_mfma_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/tools/batch_rate.py 16 8 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py "$R/gpurun_out/pmc_mfma/*/*counter_collection.csv" gemv_mfma_kernel > $R/gpurun_out/r04_mfma_pmc.txt 2>&1
rm -rf $R/gpurun_out/prof_mfma $R/gpurun_out/pmc_mfma
cat $R/gpurun_out/r04_mfma_kernel_stats.csv
cat $R/gpurun_out/r04_mfma_pmc.txt
