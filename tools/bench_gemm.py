"""Micro-benchmark of usdm_gemm on the shapes of the hot path (run on the GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan

dev = torch.device("cuda:0")


def bench(M, N, K, dtype=torch.bfloat16, act=0, residual=False, out16=True, reps=20, **kw):
    A = torch.randn(M, K, device=dev).to(dtype)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dtype)
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev) if residual else None
    o16 = torch.zeros(M, N, device=dev, dtype=torch.bfloat16) if out16 else None
    o32 = None if out16 else torch.zeros(M, N, device=dev)
    # `reps` launches recorded once and replayed as one hipGraph: eager ops.gemm costs ~15 us of Python per call,
    # which would hide any kernel shorter than that
    plan = ops.Plan()
    for _ in range(reps):
        ops.gemm(A, W, M=M, N=N, Kc=K, bias=bias, act=act, residual=R, ldr=N, out16=o16, out32=o32, plan=plan, **kw)
    gp = GraphedPlan(plan)
    for _ in range(3):
        gp.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gp.run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"M{M} N{N} K{K} {str(dtype)[6:]} act{act} res{int(residual)}: {us:8.1f} us  {2 * M * N * K / us / 1e6:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    for (M, N, K) in [(2236, 3072, 1024), (2236, 4096, 1024), (2236, 1024, 4096), (2236, 1024, 1024), (4096, 4096, 4096),
                      (584, 6144, 4096), (584, 28672, 4096), (584, 4096, 14336)]:
        bench(M, N, K)
    bench(2236, 4096, 1024, act=1)
    bench(2236, 1024, 4096, residual=True, out16=False)
    bench(499, 1280, 1280, dtype=torch.float32, out16=False)
    bench(499, 5120, 1280, dtype=torch.float32, out16=False)
    bench(499, 1280, 5120, dtype=torch.float32, out16=False)
    bench(4096, 4096, 4096, dtype=torch.float32, out16=False)
