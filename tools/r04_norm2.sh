#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python tools/norm_bench.py 2>&1 | grep -v amdgpu
python tools/vb_nfe_time.py 2>&1 | grep NFE
python tools/vb_nfe_time.py 2>&1 | grep NFE
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04_gputests_g.log 2>&1; tail -2 gpurun_out/r04_gputests_g.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nfe -- python3 $R/tools/vb_nfe_time.py > $R/gpurun_out/r04_prof_nfe.out 2>&1
f=$(find $R/gpurun_out/prof_nfe -name "*kernel_stats.csv" | head -1)
python3 $R/tools/prof_summary.py $f 8 > $R/gpurun_out/r04_vb_nfe_kernel_stats.csv
rm -rf $R/gpurun_out/prof_nfe
cat $R/gpurun_out/r04_vb_nfe_kernel_stats.csv | cut -c1-110
