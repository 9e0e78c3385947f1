"""usdm_norm at the Voicebox shape ([2236][1024] f32 in -> f32 + bf16 out), graph-replayed."""
import sys
import torch
sys.path.insert(0, ".")
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
dev = torch.device("cuda:0")
R, C = 2236, 1024
xs = [torch.randn(R, C, device=dev) for _ in range(8)]
g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
o32, o16 = torch.zeros(R, C, device=dev), torch.zeros(R, C, device=dev, dtype=torch.bfloat16)
plan = ops.Plan()
for i in range(48):
    ops.norm(xs[i % 8], g, b, rows=R, C=C, out32=o32, out16=o16, plan=plan)
gp = GraphedPlan(plan)
for _ in range(3):
    gp.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    gp.run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (5 * 48)
print(f"norm [2236][1024]: {us:.2f} us  ({R * C * 10 / us / 1e6:.2f} TB/s over 10 B/element)")
# the 7B's RMSNorm in a prefill: [S][4096] bf16 -> bf16 (HF rounding points), S = 38 and 548
for S in (38, 548):
    C = 4096
    hs = [torch.randn(S, C, device=dev).bfloat16() for _ in range(8)]
    gw = torch.randn(C, device=dev)
    xn = torch.zeros(S, C, device=dev, dtype=torch.bfloat16)
    plan = ops.Plan()
    for i in range(48):
        ops.norm(hs[i % 8], gw, None, rows=S, C=C, eps=1e-5, rms=True, round_bf16=True, out16=xn, plan=plan)
    gp = GraphedPlan(plan)
    for _ in range(3):
        gp.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gp.run()
    e1.record(); torch.cuda.synchronize()
    print(f"rmsnorm [{S}][4096] bf16: {e0.elapsed_time(e1) * 1e3 / (5 * 48):.2f} us")
