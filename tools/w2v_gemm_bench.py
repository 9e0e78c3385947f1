"""The XLS-R tokenizer's f32 GEMMs (M = 499 frames, d = 1280, ffn 5120) per tile: us per launch over cold weights.
python tools/w2v_gemm_bench.py [M]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 499
L = 12
shapes = [("qkv", 3840, 1280), ("out-proj", 1280, 1280), ("ffn1", 5120, 1280), ("ffn2", 1280, 5120)]
for name, N, K in shapes:
    Ws = [torch.randn(N, K, device=dev) * K ** -0.5 for _ in range(L)]
    X = torch.randn(M, K, device=dev)
    b = torch.randn(N, device=dev)
    Y = torch.zeros(M, N, device=dev)
    line = f"f32 M={M} {name:9s} {N:5d} x {K:5d} ({2 * M * N * K / 1e9:5.2f} GFLOP):"
    for tile in (None, 2, 5, 7, 8, 1, 6, 10, 0, 4, 11):
        if tile is not None:
            os.environ["USDM_GEMM_TILE"] = str(tile)
        else:
            os.environ.pop("USDM_GEMM_TILE", None)
        try:
            plan = ops.Plan()
            for W in Ws:
                ops.gemm(X, W, M=M, N=N, Kc=K, bias=b, out32=Y, plan=plan)
            gp = GraphedPlan(plan)
            for _ in range(3):
                gp.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gp.run()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (5 * L)
            line += f"  {'auto' if tile is None else 't' + str(tile)} {us:6.1f}"
        except Exception as e:  # noqa: BLE001
            line += f"  t{tile} n/a"
    os.environ.pop("USDM_GEMM_TILE", None)
    print(line, flush=True)
    del Ws
