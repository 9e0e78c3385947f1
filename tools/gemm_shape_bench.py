"""us per launch of usdm_gemm on a list of M,N,K shapes (bf16, plain bf16 output + bias), 8 distinct weight sets replayed in one
hipGraph (cold weights), for each tile override given in TILES (comma list; "d" = the launcher's own choice)."""
import os, sys, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from usdm_amd import ops
    from usdm_amd.graph import GraphedPlan
    dev = torch.device("cuda:0"); bf = torch.bfloat16
    for spec in sys.argv[2:]:
        p = [int(v) for v in spec.split("x")]
        M, N, K = p[:3]; sk = p[3] if len(p) > 3 else 1
        L = max(2, min(8, int(3e8 // (N * K * 2)) + 1))
        Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(bf) for _ in range(L)]
        x = torch.randn(M, K, device=dev).to(bf); b = torch.randn(N, device=dev)
        o16 = torch.zeros(M, N, device=dev, dtype=bf); o32 = torch.zeros(max(sk, 1), M, N, device=dev)
        plan = ops.Plan()
        for W in Ws:
            if sk > 1: ops.gemm(x, W, M=M, N=N, Kc=K, bias=b, out32=o32, split_k=sk, c_split_stride=M * N, plan=plan)
            else: ops.gemm(x, W, M=M, N=N, Kc=K, bias=b, out16=o16, plan=plan)
        gp = GraphedPlan(plan)
        for _ in range(3): gp.run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): gp.run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * L)
        print(f"{spec:>20s} {us:8.2f} us {2 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)
    sys.exit(0)
tiles = os.environ.get("TILES", "d,12,13").split(",")
for t in tiles:
    env = dict(os.environ)
    if t != "d": env["USDM_GEMM_TILE"] = t
    print(f"== tile {t}", flush=True)
    subprocess.run([sys.executable, __file__, "--one", *sys.argv[1:]], env=env, timeout=300)
