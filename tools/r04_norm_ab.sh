#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r04_norm_ab.log
echo "---- gamma behind the statistics for rows of 4096 (HEAD~: git stash of norm.hip is not available on the box; built from the previous object)" > $L
python tools/norm_bench.py >> $L 2>&1 || exit 1
timeout -k 10 300 python -m pytest tests/test_llm_gpu.py tests/test_voicebox_gpu.py -x -q >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu $L
