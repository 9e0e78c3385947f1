"""Per-workgroup phase timeline of one usdm_gemm launch.

Needs a library built with  USDM_EXTRA_HIPCC_FLAGS=-DUSDM_GEMM_TRACE python -m usdm_amd.build --force
(rebuild without the flag afterwards).  Usage: python tools/gemm_trace.py M N K [reps]
"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from usdm_amd import ops, _lib

M, N, K = (int(x) for x in sys.argv[1:4])
QKV = "--qkv" in sys.argv   # head-split epilogue of the Voicebox layer (N = 3 * 16 * 64)
dev = torch.device("cuda:0")
A = torch.randn(M, K, device=dev).bfloat16()
W = (torch.randn(N, K, device=dev) * K ** -0.5).bfloat16()
bias = torch.randn(N, device=dev)
out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
if QKV:
    S = M // 2; Spad = (S + 63) // 64 * 64
    q = torch.zeros(2, 16, Spad, 64, device=dev, dtype=torch.bfloat16); k = torch.zeros_like(q)
    vt = torch.zeros(2, 16, 64, Spad, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    if QKV:
        ops.gemm(A, W, M=M, N=N, Kc=K, bias=bias, qkv=dict(S=S, Spad=Spad, H=16, D=64, q=q, k=k, v=vt))
    else:
        ops.gemm(A, W, M=M, N=N, Kc=K, bias=bias, out16=out, act=1 if "--gelu" in sys.argv else 0)
torch.cuda.synchronize()
lib = _lib.lib
buf = np.zeros(8192 * 8, dtype=np.uint64)
rc = lib.usdm_dbg_gemm_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size))
assert rc == 0, rc
t = buf.reshape(8192, 8)
nwg = int((t[:, 0] != 0).sum())
t = t[:nwg].astype(np.int64)
hw = t[:, 7]
xcc = (hw >> 32) & 0xF
hid = hw & 0xFFFFFFFF
cu = (hid >> 8) & 0xF; sh = (hid >> 12) & 1; se = (hid >> 13) & 0x7
cuid = xcc * 1000 + se * 100 + sh * 10 + cu
t0 = t[:, 0].min()
ns = lambda x: x * 10  # 100 MHz ticks -> ns
print(f"workgroups {nwg}, distinct CUs {len(set(cuid.tolist()))}, kernel span {ns(t[:, 5].max() - t0) / 1e3:.2f} us")
start = ns(t[:, 0] - t0) / 1e3
print("start offset us: percentiles 0/25/50/75/100:", np.percentile(start, [0, 25, 50, 75, 100]).round(2))
names = ["prologue", "first-load->LDS", "K loop", "acc->LDS", "epilogue stores"]
for i, nm in enumerate(names):
    d = ns(t[:, i + 1] - t[:, i]) / 1e3
    print(f"  {nm:18s} median {np.median(d):6.2f} us  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}")
d = ns(t[:, 6] - t[:, 4]) / 1e3
print(f"  {'  stores issued':18s} median {np.median(d):6.2f} us  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}")
if QKV:   # the last third of the tiles along N holds V (transposed store)
    bm = 256 if nwg == 9 * 24 else (128 if nwg in (18 * 24, 18 * 48) else 64)
    tiles_n = nwg // ((M + bm - 1) // bm)
    q8, r8 = nwg // 8, nwg % 8
    bid = np.arange(nwg); xcd = bid & 7; idx = bid >> 3
    tile = np.where(xcd < r8, xcd * (q8 + 1), r8 * (q8 + 1) + (xcd - r8) * q8) + idx
    isv = (tile % tiles_n) >= (2 * tiles_n) // 3
    for nm, sel in (("Q/K tiles", ~isv), ("V tiles", isv)):
        dd = ns(t[sel, 5] - t[sel, 4]) / 1e3
        print(f"  epilogue, {nm:10s} median {np.median(dd):6.2f} us  p10 {np.percentile(dd, 10):6.2f}  p90 {np.percentile(dd, 90):6.2f}  (n={sel.sum()})")
d = ns(t[:, 5] - t[:, 0]) / 1e3
print(f"  {'whole workgroup':18s} median {np.median(d):6.2f} us  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}")
# concurrency: workgroups resident per CU at the midpoint of the first workgroup on it
import collections
per = collections.Counter(cuid.tolist())
print("workgroups per CU: min/median/max", min(per.values()), int(np.median(list(per.values()))), max(per.values()))
# histogram of start times (1 us bins)
h, _ = np.histogram(start, bins=np.arange(0, start.max() + 1.0, 1.0))
print("starts per us bin:", h.tolist())
e = ns(t[:, 5] - t0) / 1e3
h, _ = np.histogram(e, bins=np.arange(0, e.max() + 1.0, 1.0))
print("ends   per us bin:", h.tolist())

