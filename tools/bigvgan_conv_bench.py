"""BigVGAN's f32 convolutions (AMP-block Conv1d k = 3 / 7 / 11 at the six resolutions, 861 mel frames) per register-staged tile:
us per launch and TF/s.  python tools/bigvgan_conv_bench.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
dev = torch.device("cuda:0")
stages = [(3444, 768), (13776, 384), (27552, 192), (55104, 96), (110208, 48), (220416, 24)]
tot = {}
for T, C in stages:
    cp = (C + 31) // 32 * 32
    for k in (3, 7, 11):
        L = 4
        Ws = [torch.randn(C, k * cp, device=dev) * (k * C) ** -0.5 for _ in range(L)]
        X = torch.randn(T, cp, device=dev)
        b = torch.randn(C, device=dev)
        R = torch.randn(T, cp, device=dev)
        Y = torch.zeros(T, cp, device=dev)
        line = f"T={T:6d} C={C:4d} k={k:2d} ({2 * T * C * C * k / 1e9:6.2f} GFLOP):"
        for tile in (None, 2, 1, 0, 3):
            if tile is not None:
                os.environ["USDM_GEMM_TILE"] = str(tile)
            else:
                os.environ.pop("USDM_GEMM_TILE", None)
            plan = ops.Plan()
            for W in Ws:
                ops.gemm(X, W, M=T, N=C, Kc=cp, taps=k, rowsA=T, a_row_off=-(k // 2), a_row_step=1, bias=b, residual=R, ldr=cp, out32=Y, ldc=cp, plan=plan)
            gp = GraphedPlan(plan)
            for _ in range(2):
                gp.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                gp.run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (3 * L)
            line += f"  {'auto' if tile is None else 't' + str(tile)} {us:7.1f} ({2 * T * C * C * k / us / 1e6:5.1f} TF/s)"
            tot[tile] = tot.get(tile, 0.0) + us * 6      # 6 convolutions of each kernel size per stage
        os.environ.pop("USDM_GEMM_TILE", None)
        print(line, flush=True)
print("sum over the 108 AMP convolutions, ms:", {('auto' if t is None else f't{t}'): round(v / 1e3, 2) for t, v in tot.items()})
