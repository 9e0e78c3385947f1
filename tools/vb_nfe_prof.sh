# per-kernel time of the Voicebox NFE (tools/vb_nfe_time.py under rocprofv3 kernel stats); summary -> gpurun_out/vb_nfe_kernel_stats.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nfe -- python3 $R/tools/vb_nfe_time.py 20 > $R/gpurun_out/vb_nfe_prof.out 2>&1
f=$(find $R/gpurun_out/prof_nfe -name "*kernel_stats.csv" | head -1)
python3 $R/tools/prof_summary.py $f 24 > $R/gpurun_out/vb_nfe_kernel_stats.csv
rm -rf $R/gpurun_out/prof_nfe
tail -1 $R/gpurun_out/vb_nfe_prof.out
cat $R/gpurun_out/vb_nfe_kernel_stats.csv | cut -c1-200
