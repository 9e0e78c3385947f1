# rank-0-of-8 shard on one GPU: kernels + launches of a TP=8 decode step without wire time (tools/tp8_proxy.py)
cd $GRAFT_REPO_ROOT
L=gpurun_out/r02_tp8_proxy.log
: > $L
run() { echo "$1: $(env $2 python tools/tp8_proxy.py 8 2>/dev/null | tail -1)" >> $L; }
run "rccl path, combine kernel (r01 form)      " "USDM_TP_COMM=rccl USDM_ATTN_MERGE_IN_OPROJ=0"
run "rccl path, attention merged in o_proj     " "USDM_TP_COMM=rccl USDM_ATTN_MERGE_IN_OPROJ=1"
run "p2p fused epilogue, combine kernel        " "USDM_TP_COMM=p2p USDM_ATTN_MERGE_IN_OPROJ=0"
run "p2p fused epilogue, merged in o_proj      " "USDM_TP_COMM=p2p USDM_ATTN_MERGE_IN_OPROJ=1"
run "p2p split (put + reduce launch), merged   " "USDM_TP_COMM=p2p USDM_ATTN_MERGE_IN_OPROJ=1 USDM_P2P_FUSED=0"
run "p2p fused, merged, NS=4                   " "USDM_TP_COMM=p2p USDM_ATTN_MERGE_IN_OPROJ=1 USDM_DECODE_SPLITS=4"
run "p2p fused, merged, NS=16                  " "USDM_TP_COMM=p2p USDM_ATTN_MERGE_IN_OPROJ=1 USDM_DECODE_SPLITS=16"
cat $L
