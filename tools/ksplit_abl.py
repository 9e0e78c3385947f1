"""Ablation of the K-split ping-pong loop (gemm tiles 15 / 16): USDM_GEMM_ABL bits 1 = no in-loop DMA, 2 = no MFMA, 4 = no fragment
reads (results are then garbage; timing only).  Shape = the Voicebox layer GEMMs at B = 2 x 1118 rows, 24 cold weight sets, hipGraph."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
dev = torch.device("cuda:0")
bf = torch.bfloat16
R, L = 2236, 24
shapes = [("qkv 3072x1024", 3072, 1024, 0), ("w1 4096x1024", 4096, 1024, 0), ("w2 1024x4096 s3", 1024, 4096, 3)]
tiles = [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "12,15,16").split(",")]
abls = [int(t) for t in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2,3,4,5,6,7").split(",")]
for name, N, K, sk in shapes:
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(bf) for _ in range(L)]
    x = torch.randn(R, K, device=dev).to(bf)
    b = torch.randn(N, device=dev)
    o16 = torch.zeros(R, N, device=dev, dtype=bf)
    o32 = torch.zeros(max(sk, 1), R, N, device=dev)
    for tile in tiles:
        row = []
        for abl in (abls if tile >= 15 else [0]):
            os.environ["USDM_GEMM_TILE"], os.environ["USDM_GEMM_ABL"] = str(tile), str(abl)
            plan = ops.Plan()
            for W in Ws:
                if sk:
                    ops.gemm(x, W, M=R, N=N, Kc=K, bias=b, out32=o32, split_k=sk, c_split_stride=R * N, plan=plan)
                else:
                    ops.gemm(x, W, M=R, N=N, Kc=K, bias=b, out16=o16, plan=plan)
            gp = GraphedPlan(plan)
            for _ in range(3):
                gp.run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                gp.run()
            e1.record(); torch.cuda.synchronize()
            row.append(f"abl{abl}: {e0.elapsed_time(e1) * 1e3 / (5 * L):6.2f}")
        print(f"{name:18s} tile {tile:2d}  " + "  ".join(row), flush=True)
