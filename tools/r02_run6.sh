cd $GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 2 > gpurun_out/r02_bench_c.log 2>&1; echo "bench rc=$?"
tail -c 1800 gpurun_out/r02_bench_c.log
echo; echo "---- p2p proxy"
USDM_TP_COMM=p2p python tools/tp8_proxy.py 8 2>&1 | tail -6
echo "---- force-dist bench path (1 rank, p2p self-test + transport)"
python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-batched --force-dist 2>&1 | tail -c 700
