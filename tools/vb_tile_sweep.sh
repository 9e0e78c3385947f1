cd $GRAFT_REPO_ROOT
: > gpurun_out/r02_vb_tiles.log
for t in auto 4 6 11 12 13 14 15; do
  if [ $t = auto ]; then unset USDM_GEMM_TILE; else export USDM_GEMM_TILE=$t; fi
  echo "== USDM_GEMM_TILE=$t" >> gpurun_out/r02_vb_tiles.log
  timeout -k 10 120 python tools/vb_gemm_bench.py 2>&1 | grep -v "^$" | tail -9 >> gpurun_out/r02_vb_tiles.log
done
cat gpurun_out/r02_vb_tiles.log
