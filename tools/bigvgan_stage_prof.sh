# per-(kernel, grid) time of the f32 BigVGAN forward at 861 frames: which stage's launches cost what
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
BV_MODES=f32 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_bv -- python3 $R/tools/bigvgan_time.py > $R/gpurun_out/bv_prof.out 2>&1
f=$(find $R/gpurun_out/prof_bv -name "*kernel_trace.csv" | head -1)
python3 $R/tools/kdur.py $f 60 > $R/gpurun_out/bigvgan_stage_kernels.txt
rm -rf $R/gpurun_out/prof_bv
cat $R/gpurun_out/bigvgan_stage_kernels.txt
