"""Micro-benchmark of usdm_gemv on the Mistral-7B decode shapes (cold weights: a ring of buffers > 256 MiB L3)."""
import sys
import torch
sys.path.insert(0, ".")
from usdm_amd import ops

dev = torch.device("cuda:0")


def bench(N, K, act=0, norm=False, residual=False, copies=None, reps=4):
    nbytes = N * K * 2
    copies = copies or max(2, int(1.2e9 // nbytes) + 1)
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(copies)]
    x = torch.randn(K, device=dev).to(torch.bfloat16)
    g = torch.ones(K, device=dev) if norm else None
    nout = N // 2 if act == 3 else N
    r = torch.randn(nout, device=dev).to(torch.bfloat16) if residual else None
    y = torch.zeros(nout, device=dev, dtype=torch.bfloat16)
    f = lambda W: ops.gemv(W, x, N=N, K=K, norm_w=g, act=act, residual=r, y16=y)
    for W in Ws:
        f(W)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for W in Ws:
            f(W)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * copies)
    print(f"N{N} K{K} act{act} norm{int(norm)} res{int(residual)}: {us:7.2f} us  {nbytes / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    bench(6144, 4096, norm=True)
    bench(4096, 4096, residual=True)
    bench(28672, 4096, act=3, norm=True)
    bench(4096, 14336, residual=True)
    bench(42003, 4096, norm=True)
    bench(768, 4096, norm=True)      # TP=8 shards
    bench(4096, 512, residual=False)
    bench(3584, 4096, act=3, norm=True)
    bench(4096, 1792)
