"""Cycles per phase of the Voicebox attention kernel (S=1118, 2 x 16 heads, d=64).
Needs USDM_EXTRA_HIPCC_FLAGS=-DUSDM_ATTN_TRACE python -m usdm_amd.build --force."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from usdm_amd import ops, _lib
dev = torch.device("cuda:0")
Bx, nh, S = 2, 16, 1118
Spad = (S + 63) // 64 * 64
bf = torch.bfloat16
q = torch.randn(Bx, nh, Spad, 64, device=dev).to(bf); k = torch.randn(Bx, nh, Spad, 64, device=dev).to(bf)
vt = torch.randn(Bx, nh, 64, Spad, device=dev).to(bf)
o = torch.zeros(Bx * S, nh * 64, device=dev, dtype=bf)
slopes = torch.tensor([2 ** (-(i + 1) / 2) for i in range(nh)], device=dev)
kvl = torch.tensor([S, S], dtype=torch.int32, device=dev)
H = nh * 64
for _ in range(3):
    ops.attention(q, k, vt, o, mode=0, dh=64, B=Bx, Hq=nh, Hkv=nh, Sq=S, Skv=S, Skv_alloc=Spad,
                  q_strides=(nh * Spad * 64, Spad * 64, 64), k_strides=(nh * Spad * 64, Spad * 64, 64),
                  v_strides=(nh * 64 * Spad, 64 * Spad, Spad), o_strides=(S * H, H), scale=1.0, kv_len=kvl, slopes=slopes, alibi_col0_zero=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.attention(q, k, vt, o, mode=0, dh=64, B=Bx, Hq=nh, Hkv=nh, Sq=S, Skv=S, Skv_alloc=Spad,
                  q_strides=(nh * Spad * 64, Spad * 64, 64), k_strides=(nh * Spad * 64, Spad * 64, 64),
                  v_strides=(nh * 64 * Spad, 64 * Spad, Spad), o_strides=(S * H, H), scale=1.0, kv_len=kvl, slopes=slopes, alibi_col0_zero=True)
e1.record(); torch.cuda.synchronize()
print(f"kernel (eager loop, host-bound floor ~15 us): {e0.elapsed_time(e1) * 1e3 / 20:.1f} us")
if hasattr(_lib.lib, "usdm_dbg_attn_trace"):
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    assert _lib.lib.usdm_dbg_attn_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
    t = buf.reshape(4096, 8)[:9 * nh * Bx].astype(np.float64)
    names = ["prologue (first tile in LDS)", "QK^T (LDS reads + MFMA)", "softmax (VALU)", "PV (LDS reads + MFMA)", "store next tile + issue loads", "barrier wait"]
    tot = t[:, 6]
    print(f"workgroups {len(t)}, cycles per workgroup median {np.median(tot):.0f} (s_memtime ticks), tiles {np.median(t[:, 7]):.0f}")
    for i, nm in enumerate(names):
        print(f"  {nm:32s} {np.median(t[:, i]):9.0f}  ({100 * np.median(t[:, i]) / np.median(tot):4.1f} %)   per tile {np.median(t[:, i] / np.maximum(t[:, 7], 1)):7.0f}")
    # per (head) medians: workgroup index = (z * gy + y) * gx + x, head = gy - 1 - (y + gy * z) // gz (slow-heads-first order)
    lin = np.arange(len(t)) // 9
    head = nh - 1 - lin // Bx
    tick_us = 0.01   # s_memtime: 100 MHz
    for hh in range(nh):
        sel = head == hh
        print(f"  head {hh:2d}: workgroup total median {np.median(tot[sel]) * tick_us:6.2f} us   softmax+PV share {100 * np.median((t[sel, 2] + t[sel, 3]) / tot[sel]):4.1f} %")
