cd $GRAFT_REPO_ROOT
for cfg in "4 4" "5 3" "3 3"; do
  set -- $cfg
  USDM_EXTRA_HIPCC_FLAGS="-DUSDM_ENG_NL=$1 -DUSDM_ENG_NC=$2 -DUSDM_ENG_TRACE" python -m usdm_amd.build --force > /dev/null 2>&1
  echo "-- NL=$1 NC=$2"; timeout -k 10 60 python -m pytest tests/test_chain_gpu.py -x -q -k "engine_bit" 2>&1 | tail -1
  timeout -k 10 60 python tools/eng_trace.py 2>&1 | grep -v amdgpu
done
