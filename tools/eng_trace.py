"""Where the engine's waves of workgroup 0 spend their cycles (build with USDM_EXTRA_HIPCC_FLAGS=-DUSDM_ENG_TRACE)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops, _lib
from usdm_amd.llm import _pack_gate_up
dev = torch.device("cuda:0"); bf = torch.bfloat16
H, I = 4096, 14336
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s, sc: (torch.randn(*s, device=dev, generator=g) * sc).to(bf)
NLAY = 8
Ws = [_pack_gate_up(r(I, H, sc=H ** -0.5), r(I, H, sc=H ** -0.5)) for _ in range(NLAY)]
h, act = r(H, sc=1.0), torch.zeros(I, dtype=bf, device=dev)
ln = torch.ones(H, device=dev)
sync = torch.zeros(8, dtype=torch.int32, device=dev); gran = torch.zeros(3 * 8192, dtype=torch.int64, device=dev)
plan = ops.Plan()
for W in Ws:
    ops.gemv_engine([ops.gemv(W, h, N=2 * I, K=H, norm_w=ln, eps=1e-5, act=3, y16=act, only_args=True)], sync, gran, timeout_ms=300, plan=plan)
plan.run(); torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
_lib.exp().usdm_dbg_eng_trace(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); plan.run(); e1.record(); torch.cuda.synchronize()
_lib.exp().usdm_dbg_eng_trace(buf, 0)
us = e0.elapsed_time(e1) * 1e3 / NLAY
print(f"gate/up engine launch: {us:.1f} us; per launch, cycles of workgroup 0 (clock64):")
names = {0: "wait free slot", 1: "issue DMAs", 2: "wait DMAs landed"}
for lw in range(4):
    v = [buf[lw * 4 + k] / NLAY for k in range(3)]
    if any(v):
        print(f"  loader {lw}: " + ", ".join(f"{names[k]} {v[k]:9.0f}" for k in range(3)))
for cw in range(4):
    v = [buf[32 + cw * 4 + k] / NLAY for k in range(2)]
    if any(v):
        print(f"  consumer {cw}: wait full slot {v[0]:9.0f}, read+dot+reduce {v[1]:9.0f}")
