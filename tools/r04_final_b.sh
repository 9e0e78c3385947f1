#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
: > gpurun_out/r04_batch_attn_pipe_ab.log
for cfg in "256 0" "512 0" "1024 0" "512 1"; do
  timeout -k 10 200 python tools/batch_rate.py 16 256 $cfg >> gpurun_out/r04_batch_attn_pipe_ab.log 2>&1 || exit 1
done
grep -v amdgpu gpurun_out/r04_batch_attn_pipe_ab.log
python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err && tail -c 300 gpurun_out/r04_bench.json && bash tools/r04_profile.sh > gpurun_out/r04_profile_run.log 2>&1
tail -4 gpurun_out/r04_profile_run.log
