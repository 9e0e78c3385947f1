"""usdm_gemv_chain vs separate usdm_gemv launches at the 7B layer shapes, cold weights (a different layer's matrices every launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
from usdm_amd.llm import _pack_gate_up

dev = torch.device("cuda:0")
bf = torch.bfloat16
H, I, NQ, NL = 4096, 14336, 6144, 12
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s, sc: (torch.randn(*s, device=dev, generator=g) * sc).to(bf)
Ws = [dict(o=r(H, H, sc=H ** -0.5), gu=_pack_gate_up(r(I, H, sc=H ** -0.5), r(I, H, sc=H ** -0.5)), down=r(H, I, sc=I ** -0.5), qkv=r(NQ, H, sc=H ** -0.5),
           ln=torch.ones(H, device=dev)) for _ in range(NL)]
h, ao, act, qkv = r(H, sc=1.0), r(H, sc=1.0), torch.zeros(I, dtype=bf, device=dev), torch.zeros(NQ, dtype=bf, device=dev)
sync = torch.zeros(8, dtype=torch.int32, device=dev)
gran = torch.zeros(3 * 8192, dtype=torch.int64, device=dev)


def phases(W, which, **k):
    f = dict(o=lambda: ops.gemv(W["o"], ao, N=H, K=H, residual=h, y16=h, **k),
             gu=lambda: ops.gemv(W["gu"], h, N=2 * I, K=H, norm_w=W["ln"], eps=1e-5, act=3, y16=act, **k),
             down=lambda: ops.gemv(W["down"], act, N=H, K=I, residual=h, y16=h, **k),
             qkv=lambda: ops.gemv(W["qkv"], h, N=NQ, K=H, norm_w=W["ln"], eps=1e-5, y16=qkv, **k))
    return [f[w]() for w in which]


def timeit(plan, n=10):
    gp = GraphedPlan(plan)
    for _ in range(3):
        gp.run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gp.run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n / NL * 1e3


for which in (["o"], ["gu"], ["down"], ["qkv"], ["gu", "down"], ["o", "gu", "down"], ["o", "gu", "down", "qkv"]):
    sep, ch, en = ops.Plan(), ops.Plan(), ops.Plan()
    sync.zero_()          # one block per pattern (the counters are monotonic in lockstep with the generation)
    sync2 = torch.zeros(8, dtype=torch.int32, device=dev)
    for W in Ws:
        phases(W, which, plan=sep)
        ops.gemv_chain(phases(W, which, only_args=True), sync, plan=ch)
        ops.gemv_engine(phases(W, which, only_args=True), sync2, gran, timeout_ms=500, plan=en)
    try:
        t_sep, t_ch, t_en = timeit(sep), timeit(ch), timeit(en)
        print(f"{'+'.join(which):16s} separate {t_sep:7.2f} us   chain {t_ch:7.2f} us   engine {t_en:7.2f} us   (per layer, cold weights)   "
              f"err={int(sync[1])}/{int(sync2[1])}", flush=True)
    except Exception as e:
        print(which, "failed:", e, flush=True)
