cd $GRAFT_REPO_ROOT
L=gpurun_out/r02_vb_tiles5.log
: > $L
export USDM_GEMM_TILE=12
for cfg in "0 " "0 1" "1 " "2 " "4 " "3 " "6 " "7 "; do
  set -- $cfg
  export USDM_GEMM_DBG=$1
  if [ -n "$2" ]; then export VB_HOT=1; else unset VB_HOT; fi
  echo "== tile 12 DBG=$1 HOT=$2" >> $L
  timeout -k 10 120 python tools/vb_gemm_bench.py > gpurun_out/_vb.tmp 2>&1
  grep "us " gpurun_out/_vb.tmp | grep "qkv\*\|w1 " >> $L
done
cat $L
