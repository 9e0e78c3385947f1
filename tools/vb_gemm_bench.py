"""In-situ-like timing of the four GEMMs of a Voicebox layer (B=2 x 1118 rows): each kind is recorded 24 times over
24 distinct weight sets (cold weights, as in the 24-layer stack), replayed as one hipGraph, and timed per launch.
USDM_GEMM_TILE overrides the tile for the whole process."""
import sys
import torch
import os as _os; sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
dev = torch.device("cuda:0")
bf = torch.bfloat16
Bx, S, H, I, nh = 2, 1118, 1024, 4096, 16
R, Spad = Bx * S, (S + 63) // 64 * 64
L = 24
rnd = lambda *s, sc=1.0: torch.randn(*s, device=dev) * sc
Ws = dict(qkv=[rnd(3 * H, H, sc=H ** -0.5).to(bf) for _ in range(L)], wo=[rnd(H, H, sc=H ** -0.5).to(bf) for _ in range(L)],
          w1=[rnd(I, H, sc=H ** -0.5).to(bf) for _ in range(L)], w2=[rnd(H, I, sc=I ** -0.5).to(bf) for _ in range(L)])
x16, f16 = rnd(R, H).to(bf), rnd(R, I).to(bf)
h32, tmp32 = rnd(R, H), torch.zeros(R, H, device=dev)
q = torch.zeros(Bx, nh, Spad, 64, device=dev, dtype=bf); k = torch.zeros_like(q); vt = torch.zeros(Bx, nh, 64, Spad, device=dev, dtype=bf)
o16 = torch.zeros(R, I, device=dev, dtype=bf)
b3, b1, bI = rnd(3 * H), rnd(H), rnd(I)
tmp32s = torch.zeros(4, R, H, device=dev)
kinds = {
    "qkv  (N3072 K1024, head-split epilogue)": lambda W, p: ops.gemm(x16, W, M=R, N=3 * H, Kc=H, bias=b3, plan=p, qkv=dict(S=S, Spad=Spad, H=nh, D=64, q=q, k=k, v=vt)),
    "qkv* (same, plain bf16 epilogue)       ": lambda W, p: ops.gemm(x16, W, M=R, N=3 * H, Kc=H, bias=b3, out16=o16, ldc=I, plan=p),
    "wo   (N1024 K1024, +res f32 out)       ": lambda W, p: ops.gemm(x16, W, M=R, N=H, Kc=H, bias=b1, residual=h32, ldr=H, out32=tmp32, plan=p),
    "w1   (N4096 K1024, GELU bf16 out)      ": lambda W, p: ops.gemm(x16, W, M=R, N=I, Kc=H, bias=bI, act=1, out16=o16, plan=p),
    "w2   (N1024 K4096, +res f32 out)       ": lambda W, p: ops.gemm(f16, W, M=R, N=H, Kc=I, bias=b1, residual=h32, ldr=H, out32=tmp32, plan=p),
    "w2*  (same, split-K 2)                 ": lambda W, p: ops.gemm(f16, W, M=R, N=H, Kc=I, bias=b1, residual=h32, ldr=H, out32=tmp32s, split_k=2, c_split_stride=R * H, plan=p),
    "w2*  (same, split-K 4)                 ": lambda W, p: ops.gemm(f16, W, M=R, N=H, Kc=I, bias=b1, residual=h32, ldr=H, out32=tmp32s, split_k=4, c_split_stride=R * H, plan=p),
    "wo*  (N1024 K1024, split-K 2)          ": lambda W, p: ops.gemm(x16, W, M=R, N=H, Kc=H, bias=b1, residual=h32, ldr=H, out32=tmp32s, split_k=2, c_split_stride=R * H, plan=p),
}
arena2 = rnd(2, R, H).to(bf)
Ws["skip"] = [rnd(H, 2 * H, sc=(2 * H) ** -0.5).to(bf) for _ in range(L)]
skipf = lambda W, p: ops.gemm(arena2, W, M=R, N=H, Kc=H, taps=2, lda=H, rowsA=R, a_tap_stride=R * H, bias=b1, out32=tmp32, out16=o16, ldc=H, plan=p)
kinds["skip (N1024 K2x1024 two sources)      *"] = skipf
kinds["skip @tile14                          *"] = skipf
kinds["wo @tile14                            *"] = kinds["wo   (N1024 K1024, +res f32 out)       "]
wkey = ["qkv", "qkv", "wo", "w1", "w2", "w2", "w2", "wo", "skip", "skip", "wo"]
w2s3 = lambda W, p: ops.gemm(f16, W, M=R, N=H, Kc=I, bias=b1, residual=h32, ldr=H, out32=tmp32s, split_k=3, c_split_stride=R * H, plan=p)
kinds["w2*  (same, split-K 3 = the plan's)    *"] = w2s3
wkey.append("w2")
if _os.environ.get("VB_ONLY"):
    keep = [i for i, k in enumerate(kinds) if any(t in k for t in _os.environ["VB_ONLY"].split(","))]
    kinds = {k: v for i, (k, v) in enumerate(kinds.items()) if i in keep}
    wkey = [wkey[i] for i in keep]
tot = 0.0
for (name, f), wk in zip(kinds.items(), wkey):
    plan = ops.Plan()
    _os.environ.pop("USDM_GEMM_TILE", None)
    if "@tile" in name:
        _os.environ["USDM_GEMM_TILE"] = name.split("@tile")[1].split()[0]
    for W in Ws[wk]:
        f(Ws[wk][0] if _os.environ.get("VB_HOT") else W, plan)
    gp = GraphedPlan(plan)
    for _ in range(3):
        gp.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        gp.run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * L)
    Wn = Ws[wk][0]
    fl = 2 * R * Wn.shape[0] * Wn.shape[1]
    print(f"{name}: {us:7.2f} us  {fl / us / 1e6:6.1f} TF/s", flush=True)
    if "*" not in name:
        tot += us
print(f"sum of the 4 layer GEMMs: {tot:.1f} us")
