"""Per-workgroup phase timeline of one usdm_gemv launch over cold weights.
Needs a library built with  USDM_EXTRA_HIPCC_FLAGS=-DUSDM_GEMV_TRACE python -m usdm_amd.build --force."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from usdm_amd import ops, _lib
dev = torch.device("cuda:0")
lib = _lib.lib
for (N, K, act, norm, res) in ((4096, 4096, 0, False, True), (6144, 4096, 0, True, False), (4096, 14336, 0, False, True), (28672, 4096, 3, True, False))  # workgroup counts must not decrease (stale trace rows):
    copies = max(3, int(1.2e9 // (N * K * 2)))
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(copies)]
    x = torch.randn(K, device=dev).to(torch.bfloat16)
    g = torch.ones(K, device=dev) if norm else None
    nout = N // 2 if act == 3 else N
    r = torch.randn(nout, device=dev).to(torch.bfloat16) if res else None
    y = torch.zeros(nout, device=dev, dtype=torch.bfloat16)
    for W in Ws:
        ops.gemv(W, x, N=N, K=K, norm_w=g, act=act, residual=r, y16=y)
    torch.cuda.synchronize()
    buf = np.zeros(8192 * 8, dtype=np.uint64)
    assert lib.usdm_dbg_gemv_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
    t = buf.reshape(8192, 8)
    nwg = int((t[:, 0] != 0).sum()); t = t[:nwg].astype(np.int64)
    t0 = t[:, 0].min()
    us = lambda a: a * 10 / 1e3
    print(f"N{N} K{K} act{act}: {nwg} workgroups, span to last K-loop end {us(t[:, 3].max() - t0):.2f} us ({N * K * 2 / (us(t[:, 3].max() - t0)) / 1e6:.2f} TB/s)")
    print("   start skew p50/p90/max      ", np.percentile(us(t[:, 0] - t0), [50, 90, 100]).round(2))
    print("   ring issue (entry->issued)  ", np.percentile(us(t[:, 1] - t[:, 0]), [50, 90]).round(2))
    print("   x staging (+norm) incl sync ", np.percentile(us(t[:, 2] - t[:, 1]), [50, 90]).round(2))
    print("   K loop + wave reduce        ", np.percentile(us(t[:, 3] - t[:, 2]), [10, 50, 90, 100]).round(2))
    print("   workgroup end (rel. kernel) ", np.percentile(us(t[:, 3] - t0), [10, 50, 90, 100]).round(2))
    del Ws
