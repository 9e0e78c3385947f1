#!/bin/bash
# round 4: write-through output stores beyond usdm_gemm (LayerNorm kernel, Voicebox attention): A/B on one box, then the decode / e2e check
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r04_wt2_ab.log
echo "---- default build (usdm_gemm write-through, norm / attention plain)" > $L
python tools/vb_nfe_time.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 || exit 1
for flags in "-DUSDM_NORM_WT=1" "-DUSDM_ATTN_WT=1" "-DUSDM_NORM_WT=1 -DUSDM_ATTN_WT=1"; do
  touch usdm_amd/csrc/norm.hip usdm_amd/csrc/attn.hip
  USDM_EXTRA_HIPCC_FLAGS="$flags" python -m usdm_amd.build > gpurun_out/r04_build_wt2.log 2>&1 || { tail gpurun_out/r04_build_wt2.log; exit 1; }
  echo "---- rebuilt with $flags" >> $L
  python tools/vb_nfe_time.py >> $L 2>&1 && python tools/vb_nfe_time.py >> $L 2>&1 || exit 1
done
timeout -k 10 400 python -m pytest tests/test_voicebox_gpu.py tests/test_ln_fold_gpu.py tests/test_attention_gpu.py -x -q >> $L 2>&1 || { tail -20 $L; exit 1; }
grep -v amdgpu.ids $L
