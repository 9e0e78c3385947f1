#!/bin/bash
# round 4: A/B of write-through (sc1 / sc0 sc1) epilogue stores in usdm_gemm on one box (Voicebox layer GEMMs + NFE)
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r04_vbgemm_wt_ab.log
echo "---- default build (plain stores)" > $L
python tools/vb_gemm_bench.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 || exit 1
for v in 1 2; do
  touch usdm_amd/csrc/gemm.hip
  USDM_EXTRA_HIPCC_FLAGS=-DUSDM_GEMM_WT_STORES=$v python -m usdm_amd.build > gpurun_out/r04_build_wt.log 2>&1 || { tail gpurun_out/r04_build_wt.log; exit 1; }
  echo "---- rebuilt with -DUSDM_GEMM_WT_STORES=$v" >> $L
  python tools/vb_gemm_bench.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 || exit 1
  timeout -k 10 400 python -m pytest tests/test_gemm_gpu.py tests/test_voicebox_gpu.py tests/test_plan_shapes_gpu.py tests/test_ln_fold_gpu.py -x -q >> $L 2>&1 || { tail -20 $L; exit 1; }
done
grep -v amdgpu.ids $L
