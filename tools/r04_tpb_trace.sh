#!/bin/bash
# round 4: tensor-parallel batched decode tests, then the phase timeline of the matrix-core GEMV (trace build of one file)
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_tp_batch_gpu.py tests/test_batch_gpu.py tests/test_serving_gpu.py tests/test_llm_gpu.py -x -q -s > gpurun_out/r04_tpb_tests.log 2>&1 || { tail -40 gpurun_out/r04_tpb_tests.log; exit 1; }
tail -4 gpurun_out/r04_tpb_tests.log
touch usdm_amd/csrc/llm_mfma_k.hip
USDM_EXTRA_HIPCC_FLAGS=-DUSDM_MFMA_TRACE python -m usdm_amd.build > gpurun_out/r04_build_trace.log 2>&1 || { tail gpurun_out/r04_build_trace.log; exit 1; }
: > gpurun_out/r04_mfma_trace.log
for args in "4096 4096 16 --res" "4096 4096 16 --norm" "6144 4096 16 --norm" "8192 4096 16 --res" "28672 4096 16 --norm --glu" "4096 14336 16 --res" "4096 14336 16 --res --form 5" "2048 4096 16 --res"; do
  timeout -k 10 120 python tools/gemv_mfma_trace.py $args >> gpurun_out/r04_mfma_trace.log 2>&1 || { tail -20 gpurun_out/r04_mfma_trace.log; exit 1; }
done
grep -v amdgpu gpurun_out/r04_mfma_trace.log
