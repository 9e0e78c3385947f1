"""Reference point for tools/vb_gemm_bench.py: the same four Voicebox layer GEMMs (cold weights, 24 distinct sets, one hipGraph)
through torch.matmul (hipBLASLt / rocBLAS), no fused epilogue.  Not used by the product; tells what a tuned library kernel
reaches on these shapes."""
import torch
dev = torch.device("cuda:0"); bf = torch.bfloat16
R, H, I, L = 2236, 1024, 4096, 24
shapes = {"qkv": (3 * H, H), "wo": (H, H), "w1": (I, H), "w2": (H, I)}
tot = 0.0
for name, (N, K) in shapes.items():
    Ws = [torch.randn(N, K, device=dev).to(bf) for _ in range(L)]
    x = torch.randn(R, K, device=dev).to(bf)
    out = torch.empty(R, N, device=dev, dtype=bf)
    for W in Ws[:2]:
        torch.matmul(x, W.t(), out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            for W in Ws:
                torch.matmul(x, W.t(), out=out)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * L)
    print(f"{name} (M{R} N{N} K{K}): {us:7.2f} us  {2 * R * N * K / us / 1e6:6.1f} TF/s", flush=True)
    tot += us
print(f"sum of the 4 layer GEMMs (library): {tot:.1f} us")
