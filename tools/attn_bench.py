"""Voicebox attention launch time (default S=1118, 2 x 16 heads, d=64; AB_B / AB_H / AB_S override), hipGraph replay of 24 launches
over 24 distinct Q/K/V sets (as in the 24-layer stack)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
dev = torch.device("cuda:0")
Bx, nh, S = int(os.environ.get("AB_B", "2")), int(os.environ.get("AB_H", "16")), int(os.environ.get("AB_S", "1118"))
Spad = (S + 63) // 64 * 64
bf = torch.bfloat16
L = 24
qs = [torch.randn(Bx, nh, Spad, 64, device=dev).to(bf) * 0.5 for _ in range(L)]
ks = [torch.randn(Bx, nh, Spad, 64, device=dev).to(bf) * 0.5 for _ in range(L)]
vts = [torch.randn(Bx, nh, 64, Spad, device=dev).to(bf) for _ in range(L)]
H = nh * 64
slopes = torch.tensor([2 ** (-(i + 1) / 2) for i in range(nh)], device=dev)
if os.environ.get("AB_FLAT"):
    slopes = torch.full((nh,), float(os.environ["AB_FLAT"]), device=dev)     # no far-tile skipping: every key tile is computed
kvl = torch.tensor([S] * Bx, dtype=torch.int32, device=dev)
outs = {}
for v2 in os.environ.get("AB_VARIANTS", "0,1").split(","):
    os.environ["USDM_ATTN_V16"] = v2
    o = torch.zeros(Bx * S, H, device=dev, dtype=bf)
    plan = ops.Plan()
    for i in range(L):
        ops.attention(qs[i], ks[i], vts[i], o, mode=0, dh=64, B=Bx, Hq=nh, Hkv=nh, Sq=S, Skv=S, Skv_alloc=Spad,
                      q_strides=(nh * Spad * 64, Spad * 64, 64), k_strides=(nh * Spad * 64, Spad * 64, 64),
                      v_strides=(nh * 64 * Spad, 64 * Spad, Spad), o_strides=(S * H, H), scale=1.0, kv_len=kvl, slopes=slopes,
                      alibi_col0_zero=True, plan=plan)
    gp = GraphedPlan(plan)
    for _ in range(3):
        gp.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gp.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * L)
    fl = 4 * Bx * nh * S * S * 64
    outs[v2] = o.float().clone()
    print(f"USDM_ATTN_V16={v2} B={Bx} H={nh} S={S}: {us:6.2f} us per launch  ({fl / us / 1e6:5.0f} TF/s)", flush=True)
ks_ = list(outs)
for k in ks_[1:]:
    d = (outs[k] - outs[ks_[0]]).abs().max().item()
    print(f"max |out[{k}] - out[{ks_[0]}]| = {d:.3e} (max |out| {outs[ks_[0]].abs().max().item():.3f})")
