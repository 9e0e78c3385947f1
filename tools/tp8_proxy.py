"""Per-token time of the rank-0-of-8 tensor-parallel shard on ONE GPU (1-rank RCCL group): the kernel + launch + collective
call overhead of a TP=8 decode step without the xGMI wire time.  Numerically meaningless (7/8 of every sum is missing)."""
import os, sys, time
import torch
import torch.distributed as dist
import os as _os; sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
from usdm_amd import synth

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
tp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
comm = None
if os.environ.get("USDM_TP_COMM", "rccl") == "p2p":     # the peer-to-peer decode path: rank 0 exchanges with itself only (world 1)
    os.environ["USDM_P2P_PROXY"] = "1"
    from usdm_amd.p2p import P2PComm
    comm = P2PComm.in_process(1, 65, 4096)[0]
llm = synth.make_llm(dev, ctx_max=2048, tp_rank=0, tp_size=tp, group=dist.group.WORLD, tp_segments=True, p2p=comm)

def gather():   # 1-rank stand-in for the all_gather of the vocab-parallel partial arg-max
    n = llm.nparts
    llm.part_val[:n].copy_(llm.part_val_loc); llm.part_idx[:n].copy_(llm.part_idx_loc)
    llm.part_val[n:].fill_(-1e30)
llm._gather_partials = gather
ids = torch.randint(32002, 42002, (1, 600), generator=torch.Generator().manual_seed(3)).to(dev)
llm.generate(input_ids=ids, max_new_tokens=24)
torch.cuda.synchronize()
best = 0
for rep in range(2):
    t = time.perf_counter(); llm.generate(input_ids=ids, max_new_tokens=8); torch.cuda.synchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter(); llm.generate(input_ids=ids, max_new_tokens=264); torch.cuda.synchronize(); t2 = time.perf_counter() - t
    best = max(best, 256 / (t2 - t1))
print(f"tp={tp} shard on one GPU: {best:.1f} tok/s ({1e3 / best:.3f} ms/token); graph={'yes' if getattr(llm._decode, 'graph', None) is not None else 'no'} failed={getattr(llm._decode, 'failed', None)}")
dist.destroy_process_group()
