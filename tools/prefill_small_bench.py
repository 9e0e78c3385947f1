"""The four GEMMs of a SHORT 7B prefill (the 33 - 38 new tokens of rounds 2 and 3 once the prefix is reused): us per launch over cold
weights, automatic tile choice against forced tiles / split-K.  python tools/prefill_small_bench.py [M]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops
from usdm_amd._lib import ACT_SWIGLU
from usdm_amd.graph import GraphedPlan
dev, bf = torch.device("cuda:0"), torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 38
L = 16
shapes = [("qkv", 6144, 4096, 0, False), ("o (+res)", 4096, 4096, 0, True), ("gate/up swiglu", 28672, 4096, ACT_SWIGLU, False),
          ("down (+res)", 4096, 14336, 0, True)]


def timed(plan):
    gp = GraphedPlan(plan)
    for _ in range(3):
        gp.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gp.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * L)


for name, N, K, act, res in shapes:
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(bf) for _ in range(L)]
    X = torch.randn(M, K, device=dev).to(bf)
    nout = N // 2 if act else N
    H = torch.randn(M, nout, device=dev).to(bf)
    Y = torch.zeros(M, nout, device=dev, dtype=bf)
    line = f"M={M} {name:16s} {N:5d} x {K:5d} ({2 * N * K / 1e6:6.1f} MB):"
    for tile in (None, 5, 7, 8, 6, 10, 4, 9, 11, 14, 12):
        if tile is not None:
            os.environ["USDM_GEMM_TILE"] = str(tile)
        else:
            os.environ.pop("USDM_GEMM_TILE", None)
        try:
            plan = ops.Plan()
            for W in Ws:
                ops.gemm(X, W, M=M, N=N, Kc=K, act=act, round_bf16=True, residual=H if res else None, ldr=nout, out16=Y, ldc=nout, plan=plan)
            us = timed(plan)
            line += f"  {'auto' if tile is None else 't' + str(tile)} {us:6.1f}"
        except Exception as e:  # noqa: BLE001
            line += f"  t{tile} n/a"
    os.environ.pop("USDM_GEMM_TILE", None)
    print(line, flush=True)
    if res:      # split-K: f32 partials (the sum + residual + rounding would be one more small launch)
        for sk in ():
            P = torch.zeros(sk, M, N, device=dev)
            line = f"      split-K {sk}:"
            for tile in (None, 5, 6, 14):
                if tile is not None:
                    os.environ["USDM_GEMM_TILE"] = str(tile)
                else:
                    os.environ.pop("USDM_GEMM_TILE", None)
                try:
                    plan = ops.Plan()
                    for W in Ws:
                        ops.gemm(X, W, M=M, N=N, Kc=K, out32=P, split_k=sk, c_split_stride=M * N, plan=plan)
                    line += f"  {'auto' if tile is None else 't' + str(tile)} {timed(plan):6.1f}"
                except Exception as e:  # noqa: BLE001
                    line += f"  t{tile} n/a ({str(e)[:40]})"
            os.environ.pop("USDM_GEMM_TILE", None)
            print(line, flush=True)
    del Ws
