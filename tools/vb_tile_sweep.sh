cd $GRAFT_REPO_ROOT
: > gpurun_out/r02_vb_tiles3.log
for t in 12 13 14 15; do
  export USDM_GEMM_TILE=$t
  echo "== USDM_GEMM_TILE=$t" >> gpurun_out/r02_vb_tiles3.log
  timeout -k 10 120 python tools/vb_gemm_bench.py 2>&1 | grep -v "^$" | grep -v amdgpu | tail -9 >> gpurun_out/r02_vb_tiles3.log
done
cat gpurun_out/r02_vb_tiles3.log
