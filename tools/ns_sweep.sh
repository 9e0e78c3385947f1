set -e
cd $GRAFT_REPO_ROOT
for cfg in "0 32" "1 8" "1 12" "1 16" "1 32"; do
  set -- $cfg
  echo "merge=$1 NS=$2: $(USDM_ATTN_MERGE_IN_OPROJ=$1 USDM_DECODE_SPLITS=$2 python tools/decode_rate.py 256 600 2>/dev/null | tail -1)" >> gpurun_out/r02_ns_sweep.log
done
cd /tmp && export TMPDIR=/tmp
USDM_ATTN_MERGE_IN_OPROJ=1 USDM_DECODE_SPLITS=8 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_m8 -- python3 $GRAFT_REPO_ROOT/tools/decode_rate.py 64 600 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_m8 -name "*kernel_trace.csv" | head -1)
python tools/kdur.py $f 12 >> gpurun_out/r02_ns_sweep.log
cat gpurun_out/r02_ns_sweep.log
