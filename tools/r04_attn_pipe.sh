#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_batch_gpu.py tests/test_serving_gpu.py tests/test_tp_batch_gpu.py tests/test_llm_gpu.py -x -q > gpurun_out/r04_attn_pipe_tests.log 2>&1 || { tail -30 gpurun_out/r04_attn_pipe_tests.log; exit 1; }
tail -2 gpurun_out/r04_attn_pipe_tests.log
: > gpurun_out/r04_batch_rate_pipe.log
for b in 16 8; do timeout -k 10 300 python tools/batch_rate.py $b 256 >> gpurun_out/r04_batch_rate_pipe.log 2>&1 || exit 1; done
grep -v amdgpu gpurun_out/r04_batch_rate_pipe.log
