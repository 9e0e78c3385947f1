cd $GRAFT_REPO_ROOT
for t in 12 13; do
export USDM_GEMM_TILE=$t
for d in 0 1 2 3; do
  export USDM_GEMM_DBG=$d
  timeout -k 10 120 python tools/gemm_trace.py 2236 $1 1024 > gpurun_out/_tr.tmp 2>&1
  echo "tile $t DBG=$d: $(grep 'K loop' gpurun_out/_tr.tmp) | $(grep 'kernel span' gpurun_out/_tr.tmp)"
done
done > gpurun_out/pp_dbg_trace.log 2>&1
unset USDM_GEMM_DBG
export USDM_GEMM_TILE=12
timeout -k 10 120 python tools/gemm_trace.py 2236 3072 1024 > gpurun_out/_tr12.tmp 2>&1
export USDM_GEMM_TILE=13
timeout -k 10 120 python tools/gemm_trace.py 2236 4096 1024 > gpurun_out/_tr13.tmp 2>&1
cat gpurun_out/pp_dbg_trace.log; grep -v amdgpu gpurun_out/_tr12.tmp | head -9;  grep -v amdgpu gpurun_out/_tr13.tmp | head -9
