"""Per-shape timing of the decode projections at batch nb (python tools/gemv_mfma_bench.py [nb] [form]): the 7B's four per-layer
shapes + lm_head over 24 cold weight sets (8 for lm_head), one hipGraph per shape, us per launch and weight-stream GB/s."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
dev = torch.device("cuda:0")
bf = torch.bfloat16
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
form = int(sys.argv[2]) if len(sys.argv) > 2 else 0
shapes = [("qkv   6144 x 4096 (norm)", 6144, 4096, 0, True, False, 24), ("o     4096 x 4096 (+res)", 4096, 4096, 0, False, True, 24),
          ("gu   28672 x 4096 (norm, swiglu)", 28672, 4096, 3, True, False, 24), ("down  4096 x 14336 (+res)", 4096, 14336, 0, False, True, 24),
          ("lm_head 42003 x 4096 (norm)", 42003, 4096, 0, True, False, 8)]
if os.environ.get("EXTRA") == "1":      # what the small launches pay for: the RMSNorm prologue (same shape with / without), tiles per workgroup
    shapes = [("qkv   6144 x 4096 (norm)", 6144, 4096, 0, True, False, 24), ("qkv   6144 x 4096 (no norm)", 6144, 4096, 0, False, False, 24),
              ("o     4096 x 4096 (+res)", 4096, 4096, 0, False, True, 24), ("o     4096 x 4096 (norm)", 4096, 4096, 0, True, False, 24),
              ("o2    8192 x 4096 (+res)", 8192, 4096, 0, False, True, 24), ("o4   16384 x 4096 (+res)", 16384, 4096, 0, False, True, 12),
              ("half  2048 x 4096 (+res)", 2048, 4096, 0, False, True, 24)]
tot = 0.0
for name, N, K, act, norm, res, L in shapes:
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(bf) for _ in range(L)]
    nout = N // 2 if act == 3 else N
    X = torch.randn(nb, K, device=dev).to(bf)
    g = torch.ones(K, device=dev) if norm else None
    R = torch.randn(nb, nout, device=dev).to(bf) if res else None
    Y = torch.zeros(nb, nout, device=dev, dtype=bf)
    lm = name.startswith("lm_head")
    n = ops.gemv_nblocks(N)
    pv, pi = torch.zeros(nb, n, device=dev), torch.zeros(nb, n, dtype=torch.int32, device=dev)
    plan = ops.Plan()
    ksf = ops.gemv_batch_ks_floats(N, K) if (nb > 4 or form in (1,)) and not lm else 0
    ks = (torch.zeros(ksf, device=dev), torch.zeros(-(-N // 16), dtype=torch.int32, device=dev)) if ksf and form != 5 else None
    for W in Ws:
        if nb == 1 and form == 0:
            if lm:
                ops.gemv(W, X[0], N=N, K=K, norm_w=g, part_val=pv[0], part_idx=pi[0], plan=plan)
            else:
                ops.gemv(W, X[0], N=N, K=K, norm_w=g, act=act, residual=R[0] if res else None, y16=Y[0], plan=plan)
        elif lm:
            ops.gemv_batch(W, X, nb=nb, N=N, K=K, x_bs=K, part_bs=n, norm_w=g, part_val=pv, part_idx=pi, form=form, plan=plan)
        else:
            ops.gemv_batch(W, X, nb=nb, N=N, K=K, x_bs=K, y_bs=nout, res_bs=nout, norm_w=g, act=act, residual=R, y16=Y, form=form, ks=ks, plan=plan)
    gp = GraphedPlan(plan)
    for _ in range(3):
        gp.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gp.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * L)
    print(f"nb={nb} form={form} {name:34s} {us:7.2f} us  {2 * N * K / us / 1e3:7.1f} GB/s", flush=True)
    tot += us * (1 if lm else 32)
    del Ws
print(f"projections of one decode step (32 layers + lm_head): {tot / 1e3:.3f} ms")
