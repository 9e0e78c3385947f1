#!/bin/bash
# round 4: BigVGAN with / without write-through stores in the snake kernel and in usdm_gemm (f32 path), then the full GPU suite + bench
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r04_bigvgan_wt_ab.log
echo "---- default build (usdm_gemm + usdm_norm write-through)" > $L
BV_MODES=f32 python tools/bigvgan_time.py >> $L 2>&1 || exit 1
touch usdm_amd/csrc/bigvgan_k.hip
USDM_EXTRA_HIPCC_FLAGS="-DUSDM_SNAKE_WT=1" python -m usdm_amd.build > gpurun_out/r04_build_wt3.log 2>&1 || exit 1
echo "---- rebuilt with -DUSDM_SNAKE_WT=1" >> $L
BV_MODES=f32 python tools/bigvgan_time.py >> $L 2>&1 || exit 1
touch usdm_amd/csrc/bigvgan_k.hip usdm_amd/csrc/gemm.hip
USDM_EXTRA_HIPCC_FLAGS="-DUSDM_GEMM_WT_STORES=0" python -m usdm_amd.build > gpurun_out/r04_build_wt3.log 2>&1 || exit 1
echo "---- rebuilt with -DUSDM_GEMM_WT_STORES=0 (plain stores everywhere but usdm_norm)" >> $L
BV_MODES=f32 python tools/bigvgan_time.py >> $L 2>&1 || exit 1
grep -v amdgpu.ids $L
touch usdm_amd/csrc/bigvgan_k.hip usdm_amd/csrc/gemm.hip
python -m usdm_amd.build > gpurun_out/r04_build_wt3.log 2>&1 || exit 1
timeout -k 10 800 python -m pytest tests -q -m gpu > gpurun_out/r04_gputests_d.log 2>&1; tail -3 gpurun_out/r04_gputests_d.log
