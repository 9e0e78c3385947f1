"""lm_head GEMV (42003 x 4096) with the three ban masks of the reference's rounds: time per launch (graph-replayed)."""
import sys
import torch
sys.path.insert(0, ".")
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
dev = torch.device("cuda:0")
V, K = 42003, 4096
Ws = [(torch.randn(V, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(4)]
x = torch.randn(K, device=dev).to(torch.bfloat16)
g = torch.ones(K, device=dev)
n = ops.gemv_nblocks(V)
pv, pi = torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
masks = {"none": [], "unit->text (ban 32000..42002)": range(32000, 42003), "text->unit (ban 0..32001 but 28705)": [i for i in range(32002) if i != 28705]}
for name, ids in masks.items():
    ban = torch.zeros(V, dtype=torch.uint8)
    ban[list(ids)] = 1
    ban = ban.to(dev)
    plan = ops.Plan()
    for W in Ws * 4:
        ops.gemv(W, x, N=V, K=K, norm_w=g, ban=ban, part_val=pv, part_idx=pi, plan=plan)
    gp = GraphedPlan(plan)
    for _ in range(3):
        gp.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gp.run(); e1.record(); torch.cuda.synchronize()
    print(f"{name:40s}: {e0.elapsed_time(e1) * 1e3 / 16:6.2f} us per launch, allowed rows {int((ban == 0).sum())}")
