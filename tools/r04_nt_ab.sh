#!/bin/bash
# round 4: A/B of non-temporal epilogue stores in usdm_gemm on one box (Voicebox layer GEMMs + NFE)
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r04_vbgemm_nt_ab.log
echo "---- default build (plain stores)" > $L
python tools/vb_gemm_bench.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 || exit 1
touch usdm_amd/csrc/gemm.hip
USDM_EXTRA_HIPCC_FLAGS=-DUSDM_GEMM_NT_STORES=1 python -m usdm_amd.build > gpurun_out/r04_build_nt.log 2>&1 || { tail gpurun_out/r04_build_nt.log; exit 1; }
echo "---- rebuilt with -DUSDM_GEMM_NT_STORES=1" >> $L
python tools/vb_gemm_bench.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 || exit 1
timeout -k 10 300 python -m pytest tests/test_gemm_gpu.py tests/test_voicebox_gpu.py -x -q >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L
