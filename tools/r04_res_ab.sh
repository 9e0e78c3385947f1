#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r04_res_prefetch_ab.log
echo "---- residual requested behind the first ring (default)" > $L
python tools/decode_rate.py 384 >> $L 2>&1 && DECODE_BAN=t2u python tools/decode_rate.py 384 >> $L 2>&1 || exit 1
touch usdm_amd/csrc/llm_k.hip
USDM_EXTRA_HIPCC_FLAGS=-DUSDM_GEMV_RES_PREFETCH=0 python -m usdm_amd.build > gpurun_out/r04_build_res.log 2>&1 || exit 1
echo "---- rebuilt with -DUSDM_GEMV_RES_PREFETCH=0 (residual read in the epilogue, as in round 3)" >> $L
python tools/decode_rate.py 384 >> $L 2>&1 && DECODE_BAN=t2u python tools/decode_rate.py 384 >> $L 2>&1 || exit 1
touch usdm_amd/csrc/llm_k.hip
python -m usdm_amd.build > gpurun_out/r04_build_res.log 2>&1 || exit 1
echo "---- default again" >> $L
python tools/decode_rate.py 384 >> $L 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_llm_gpu.py tests/test_gemv_gpu.py tests/test_p2p_gpu.py tests/test_batch_gpu.py -x -q >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu $L
