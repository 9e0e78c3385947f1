set -e
cd $GRAFT_REPO_ROOT
USDM_EXTRA_HIPCC_FLAGS=-DUSDM_GEMM_TRACE python -m usdm_amd.build --force > gpurun_out/r04_trace_build.log 2>&1
for t in 12 15; do
  echo "=== tile $t, 2236 x 3072 x 1024 plain bf16"; USDM_GEMM_TILE=$t python tools/gemm_trace.py 2236 3072 1024
done
echo "=== tile 16 w1 gelu"; USDM_GEMM_TILE=16 python tools/gemm_trace.py 2236 4096 1024 --gelu
