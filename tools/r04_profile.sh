# Round-4 profile collection on the GPU box: kernel stats of the bench command, PMC (HBM bytes) of the decode GEMV.
# Usage: bash tools/r04_profile.sh   (writes under gpurun_out/; the summaries are copied into profiles/ afterwards)
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-batched > $R/gpurun_out/r04_prof_bench.out 2>&1
f=$(find $R/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1)
python3 $R/tools/prof_summary.py $f 40 > $R/gpurun_out/r04_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/decode_only.py 16 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/decode_only.py 16 > /dev/null 2>&1
{
  echo "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no tracing), python3 tools/decode_only.py 16, kernel filter gemv_kernel"
  python3 $R/tools/pmc_summary.py "$R/gpurun_out/pmc_fetch/*/*counter_collection.csv" gemv_kernel
  python3 $R/tools/pmc_summary.py "$R/gpurun_out/pmc_write/*/*counter_collection.csv" gemv_kernel
} > $R/gpurun_out/r04_gemv_pmc.txt 2>&1
# algorithmic bytes per launch of decode_only's plain mask (all 42 003 lm_head rows): 14.302 GB / 129 launches
python3 $R/tools/make_pmc_json.py $R/gpurun_out/r04_gemv_pmc.txt $R/gpurun_out/r04_gemv_pmc.json 110870000
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nfe -- python3 $R/tools/vb_nfe_time.py > $R/gpurun_out/r04_prof_nfe.out 2>&1
f=$(find $R/gpurun_out/prof_nfe -name "*kernel_stats.csv" | head -1)
python3 $R/tools/prof_summary.py $f 14 > $R/gpurun_out/r04_vb_nfe_kernel_stats.csv
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/prof_nfe
tail -3 $R/gpurun_out/r04_prof_bench.out | cut -c1-600
head -30 $R/gpurun_out/r04_bench_kernel_stats.csv
cat $R/gpurun_out/r04_gemv_pmc.txt
grep NFE $R/gpurun_out/r04_prof_nfe.out; cat $R/gpurun_out/r04_vb_nfe_kernel_stats.csv
