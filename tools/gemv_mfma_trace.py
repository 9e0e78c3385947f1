"""Per-workgroup phase timeline of one matrix-core usdm_gemv_batch launch (wave 0 of every workgroup).

Needs a library built with  USDM_EXTRA_HIPCC_FLAGS=-DUSDM_MFMA_TRACE python -m usdm_amd.build --force  (rebuild without the flag
afterwards).  Usage: python tools/gemv_mfma_trace.py N K [nb] [--norm] [--res] [--glu] [--form F]
"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from usdm_amd import ops, _lib

N, K = int(sys.argv[1]), int(sys.argv[2])
nb = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else 16
norm, res, glu = "--norm" in sys.argv, "--res" in sys.argv, "--glu" in sys.argv
form = int(sys.argv[sys.argv.index("--form") + 1]) if "--form" in sys.argv else 1
dev, bf = torch.device("cuda:0"), torch.bfloat16
L = 12
Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(bf) for _ in range(L)]      # cold weights for every launch
X = torch.randn(nb, K, device=dev).to(bf)
g = torch.ones(K, device=dev) if norm else None
nout = N // 2 if glu else N
R = torch.randn(nb, nout, device=dev).to(bf) if res else None
Y = torch.zeros(nb, nout, device=dev, dtype=bf)
ksf = ops.gemv_batch_ks_floats(N, K)
ks = (torch.zeros(ksf, device=dev), torch.zeros(-(-N // 16), dtype=torch.int32, device=dev)) if ksf and form != 5 else None
for W in Ws:
    ops.gemv_batch(W, X, nb=nb, N=N, K=K, x_bs=K, y_bs=nout, res_bs=nout, norm_w=g, act=3 if glu else 0, residual=R, y16=Y, form=form, ks=ks)
torch.cuda.synchronize()
buf = np.zeros(512 * 8, dtype=np.uint64)
rc = _lib.lib.usdm_dbg_mfma_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size))
assert rc == 0, rc
t = buf.reshape(512, 8)
nwg = int((t[:, 0] != 0).sum())
t = t[:nwg].astype(np.int64)
t0 = t[:, 0].min()
us = lambda x: x * 10 / 1e3      # 100 MHz ticks
print(f"N {N} K {K} nb {nb} norm {norm} res {res} glu {glu} form {form}: workgroups {nwg}, kernel span {us(t[:, 6].max() - t0):.2f} us")
print("  start offset us, percentiles 0/50/100:", np.percentile(us(t[:, 0] - t0), [0, 50, 100]).round(2))
names = [("loads issued", 0, 1), ("activations ready (+ RMSNorm)", 1, 2), ("first tile multiplied", 2, 3), ("rest of the stream", 3, 4),
         ("barrier of the flush", 4, 5), ("reduce + (merge) + epilogue, stores drained", 5, 6), ("whole workgroup", 0, 6)]
for nm, a, b in names:
    d = us(t[:, b] - t[:, a])
    print(f"  {nm:46s} median {np.median(d):6.2f} us  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}")
