#!/bin/bash
# round 4: K-split down_proj validation + timing, then the A/B of the 16-byte epilogue stores on one box
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_batch_gpu.py -x -q -k "k_split or matrix" > gpurun_out/r04_ks_tests.log 2>&1 || { tail -30 gpurun_out/r04_ks_tests.log; exit 1; }
tail -3 gpurun_out/r04_ks_tests.log
timeout -k 10 200 python tools/gemv_mfma_bench.py 16 0 > gpurun_out/r04_gemv_mfma_bench_ks.log 2>&1 || exit 1
timeout -k 10 200 python tools/gemv_mfma_bench.py 16 5 >> gpurun_out/r04_gemv_mfma_bench_ks.log 2>&1 || exit 1
timeout -k 10 200 python tools/gemv_mfma_bench.py 8 0 >> gpurun_out/r04_gemv_mfma_bench_ks.log 2>&1 || exit 1
cat gpurun_out/r04_gemv_mfma_bench_ks.log
for b in 16 8; do timeout -k 10 300 python tools/batch_rate.py $b 256 >> gpurun_out/r04_batch_rate_ks.log 2>&1 || exit 1; done
cat gpurun_out/r04_batch_rate_ks.log
# A/B of the wide stores
python tools/vb_gemm_bench.py > gpurun_out/r04_vbgemm_wide_ab.log 2>&1 && python tools/vb_nfe_time.py >> gpurun_out/r04_vbgemm_wide_ab.log 2>&1 || exit 1
USDM_EXTRA_HIPCC_FLAGS=-DUSDM_GEMM_WIDE16=0 python -m usdm_amd.build --force > gpurun_out/r04_build_nowide.log 2>&1 || exit 1
echo "---- rebuilt with -DUSDM_GEMM_WIDE16=0" >> gpurun_out/r04_vbgemm_wide_ab.log
python tools/vb_gemm_bench.py >> gpurun_out/r04_vbgemm_wide_ab.log 2>&1 && python tools/vb_nfe_time.py >> gpurun_out/r04_vbgemm_wide_ab.log 2>&1 || exit 1
grep -v amdgpu.ids gpurun_out/r04_vbgemm_wide_ab.log
