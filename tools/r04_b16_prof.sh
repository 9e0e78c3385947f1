#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b16 -- python3 $R/tools/batch_rate.py 16 64 > $R/gpurun_out/r04_b16_prof.out 2>&1
f=$(find $R/gpurun_out/prof_b16 -name "*kernel_stats.csv" | head -1)
python3 $R/tools/prof_summary.py $f 14 > $R/gpurun_out/r04_b16_kernel_stats.csv
rm -rf $R/gpurun_out/prof_b16
cat $R/gpurun_out/r04_b16_kernel_stats.csv
