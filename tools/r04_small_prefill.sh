#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gemm_gpu.py tests/test_llm_gpu.py tests/test_plan_shapes_gpu.py tests/test_voicebox_gpu.py tests/test_serving_gpu.py -x -q > gpurun_out/r04_small_prefill_tests.log 2>&1 || { tail -30 gpurun_out/r04_small_prefill_tests.log; exit 1; }
tail -3 gpurun_out/r04_small_prefill_tests.log
python bench.py --no-cpu-baseline --no-batched > gpurun_out/r04_bench_sp.json 2> gpurun_out/r04_bench_sp.err || exit 1
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_sp.json').read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["stage_ms"])
PY
