#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r04_xfirst_ab.log
echo "---- input piece + RMSNorm weights requested before the weight ring (default)" > $L
python tools/decode_rate.py 384 >> $L 2>&1 && DECODE_BAN=t2u python tools/decode_rate.py 384 >> $L 2>&1 || exit 1
python tools/gemv_mfma_bench.py 1 0 >> $L 2>&1 || exit 1
touch usdm_amd/csrc/llm_k.hip
USDM_EXTRA_HIPCC_FLAGS=-DUSDM_GEMV_X_FIRST=0 python -m usdm_amd.build > gpurun_out/r04_build_xf.log 2>&1 || exit 1
echo "---- rebuilt with -DUSDM_GEMV_X_FIRST=0 (behind the ring, read twice: round 3)" >> $L
python tools/decode_rate.py 384 >> $L 2>&1 && DECODE_BAN=t2u python tools/decode_rate.py 384 >> $L 2>&1 || exit 1
python tools/gemv_mfma_bench.py 1 0 >> $L 2>&1 || exit 1
touch usdm_amd/csrc/llm_k.hip
python -m usdm_amd.build > gpurun_out/r04_build_xf.log 2>&1 || exit 1
echo "---- default again" >> $L
python tools/decode_rate.py 384 >> $L 2>&1 || exit 1
grep -v amdgpu $L
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04_gputests_i.log 2>&1; tail -2 gpurun_out/r04_gputests_i.log
