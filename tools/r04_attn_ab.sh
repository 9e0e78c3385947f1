#!/bin/bash
# round 4: decode attention of the 16-sequence step: context-split workgroup target x merge placement
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r04_batch_attn_ab.log
for cfg in "512 0" "1024 0" "2048 0" "512 1" "1024 1" "2048 1"; do
  timeout -k 10 200 python tools/batch_rate.py 16 256 $cfg >> gpurun_out/r04_batch_attn_ab.log 2>&1 || { tail -20 gpurun_out/r04_batch_attn_ab.log; exit 1; }
done
grep -v amdgpu gpurun_out/r04_batch_attn_ab.log
EXTRA=1 timeout -k 10 200 python tools/gemv_mfma_bench.py 16 0 > gpurun_out/r04_gemv_mfma_extra.log 2>&1 || exit 1
EXTRA=1 timeout -k 10 200 python tools/gemv_mfma_bench.py 1 0 >> gpurun_out/r04_gemv_mfma_extra.log 2>&1 || exit 1
grep -v amdgpu gpurun_out/r04_gemv_mfma_extra.log
