"""Phase cycles of attn_vb_kernel (needs USDM_EXTRA_HIPCC_FLAGS=-DUSDM_ATTN_TRACE python -m usdm_amd.build --force)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops, _lib
dev = torch.device("cuda:0")
Bx, nh, S = int(os.environ.get("AB_B", "2")), 16, 1118
Spad = (S + 63) // 64 * 64
bf = torch.bfloat16
q = torch.randn(Bx, nh, Spad, 64, device=dev).to(bf) * 0.5; k = torch.randn(Bx, nh, Spad, 64, device=dev).to(bf) * 0.5
vt = torch.randn(Bx, nh, 64, Spad, device=dev).to(bf)
H = nh * 64
o = torch.zeros(Bx * S, H, device=dev, dtype=bf)
slopes = torch.tensor([2 ** (-(i + 1) / 2) for i in range(nh)], device=dev)
kvl = torch.tensor([S] * Bx, dtype=torch.int32, device=dev)
for _ in range(5):
    ops.attention(q, k, vt, o, mode=0, dh=64, B=Bx, Hq=nh, Hkv=nh, Sq=S, Skv=S, Skv_alloc=Spad, q_strides=(nh * Spad * 64, Spad * 64, 64),
                  k_strides=(nh * Spad * 64, Spad * 64, 64), v_strides=(nh * 64 * Spad, 64 * Spad, Spad), o_strides=(S * H, H), scale=1.0,
                  kv_len=kvl, slopes=slopes, alibi_col0_zero=True)
torch.cuda.synchronize()
n = 9 * nh * Bx
buf = np.zeros(4096 * 8, dtype=np.uint64)
assert _lib.lib.usdm_dbg_attn_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
t = buf.reshape(4096, 8)[:n].astype(np.float64)
names = ["barrier wait", "steady-state blocks", "other arm + top of step", "vmcnt wait", "n steady steps", "prologue"]
print(f"workgroups {n}; median total cycles {np.median(t[:, 6]):.0f}, wall {np.median(t[:, 7]) / 100:.2f} us -> clock {np.median(t[:, 6] / t[:, 7]) * 100:.0f} MHz")
for i, nm in enumerate(names):
    print(f"  {nm:28s} median {np.median(t[:, i]):9.0f}   mean {t[:, i].mean():9.0f}")
print("  cycles per steady step: %.0f" % np.median(t[:, 1] / np.maximum(t[:, 4], 1)))
lin = np.arange(n) // 9
head = nh - 1 - lin // Bx
for hh in (0, 5, 10, 15):
    sel = head == hh
    print(f"  head {hh:2d}: total {np.median(t[sel, 6]):8.0f} cycles, wall {np.median(t[sel, 7]) / 100:6.2f} us, steady steps {np.median(t[sel, 4]):.0f}")
