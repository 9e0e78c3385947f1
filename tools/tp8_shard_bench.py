"""Per-launch times of the decode step's kernels at the TENSOR-PARALLEL shard shapes (rank 0 of tp, default 8) on one GPU:
every kind recorded once per layer over 32 distinct (cold) weight sets, replayed as one hipGraph.  No exchange, no wire time:
this is the kernel + launch-boundary floor of a TP decode layer.  python tools/tp8_shard_bench.py [tp] [ctx]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import ops
from usdm_amd.graph import GraphedPlan
from usdm_amd.llm import _pack_gate_up

dev = torch.device("cuda:0")
bf = torch.bfloat16
tp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 600
H, I, Hq, Hkv, d, NL, CTX = 4096, 14336 // tp, 32 // tp, 8 // tp, 128, 32, 2048
NQ = (Hq + 2 * Hkv) * d
g = torch.Generator(device=dev).manual_seed(1)
r = lambda *s, sc: (torch.randn(*s, device=dev, generator=g) * sc).to(bf)
Ws = [dict(qkv=r(NQ, H, sc=H ** -0.5), o=r(H, Hq * d, sc=H ** -0.5), gu=_pack_gate_up(r(I, H, sc=H ** -0.5), r(I, H, sc=H ** -0.5)),
           down=r(H, I, sc=I ** -0.5), ln=torch.ones(H, device=dev),
           kc=r(Hkv, CTX, d, sc=1.0), vc=r(Hkv, CTX, d, sc=1.0)) for _ in range(NL)]
h, ao, act, qkv = r(H, sc=1.0), r(Hq * d, sc=1.0), torch.zeros(I, dtype=bf, device=dev), r(NQ, sc=1.0)
pos = torch.full((1,), ctx, dtype=torch.int32, device=dev)
inv = 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
fr = torch.arange(CTX).float()[:, None] * inv[None]
cos, sin = fr.cos().to(bf).to(dev), fr.sin().to(bf).to(dev)


def scratch(NS):
    f = torch.float32
    return (torch.zeros(Hq * NS, dtype=f, device=dev), torch.zeros(Hq * NS, dtype=f, device=dev), torch.zeros(Hq * NS * d, dtype=f, device=dev))


def attn(W, NS, plan, sc):
    ops.attn_decode(qkv, pos, cos, sin, W["kc"], W["vc"], *sc, ao, Hq=Hq, Hkv=Hkv, ctx_max=CTX, NS=NS, scale=d ** -0.5, plan=plan)


kinds = {
    "qkv   (N%d K4096, RMSNorm prologue)" % NQ: lambda W, p: ops.gemv(W["qkv"], h, N=NQ, K=H, norm_w=W["ln"], eps=1e-5, y16=qkv, plan=p),
    "o     (N4096 K%d, +residual)" % (Hq * d): lambda W, p: ops.gemv(W["o"], ao, N=H, K=Hq * d, residual=h, y16=h, plan=p),
    "gu    (N%d K4096, RMSNorm + SwiGLU)" % (2 * I): lambda W, p: ops.gemv(W["gu"], h, N=2 * I, K=H, norm_w=W["ln"], eps=1e-5, act=3, y16=act, plan=p),
    "down  (N4096 K%d, +residual)" % I: lambda W, p: ops.gemv(W["down"], act, N=H, K=I, residual=h, y16=h, plan=p),
}


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


print(f"tp = {tp}: shard of rank 0, context {ctx}; us per launch (hipGraph of {NL} launches over {NL} cold weight sets)")
tot = 0.0
for name, f in kinds.items():
    plan = ops.Plan()
    for W in Ws:
        f(W, plan)
    t = timeit(plan)
    Wn = [k for k in ("qkv", "o", "gu", "down") if name.startswith(k)][0]
    mb = Ws[0][Wn].numel() * 2 / 1e6
    print(f"  {name:42s} {t:6.2f} us   {mb:6.1f} MB -> {mb / t:5.2f} TB/s")
    tot += t
print(f"  sum of the four projections: {tot:.2f} us per layer")
for NS in (32, 16, 8, 4, 1):
    plan = ops.Plan()
    sc = scratch(max(NS, 1))
    for W in Ws:
        attn(W, NS, plan, sc)
    t = timeit(plan)
    print(f"  attention decode, NS = {NS:2d} ({'one launch, 16 waves per kv head' if NS == 1 else 'split + combine launch'}): {t:6.2f} us")
# the whole layer as the decode step launches it (4 projections + attention), default NS
for NS in (32, 8, 1):
    plan = ops.Plan()
    sc = scratch(max(NS, 1))
    for W in Ws:
        kinds[list(kinds)[0]](W, plan)
        attn(W, NS, plan, sc)
        for k in list(kinds)[1:]:
            kinds[k](W, plan)
    t = timeit(plan)
    nl = 4 + (1 if NS == 1 else 2)
    print(f"  whole layer, NS = {NS:2d}: {t:6.2f} us per layer ({nl} launches) -> {t * NL / 1e3:.3f} ms per token over {NL} layers (no lm_head, no exchange)")
# persistent chains where the kernels accept the shard shapes (K >= 4096 and a multiple of 512: qkv and gate/up only; the row-parallel
# shards o (K = 512) and down (K = 1792) are outside both kernels' lane partition)
sync = torch.zeros(2, 8, dtype=torch.int32, device=dev)
gran = torch.zeros(3 * 8192, dtype=torch.int64, device=dev)
for which, key in ((("gu",), 2), (("qkv",), 0)):
    name = list(kinds)[key]
    mk = lambda W, **k: ops.gemv(W["gu"], h, N=2 * I, K=H, norm_w=W["ln"], eps=1e-5, act=3, y16=act, **k) if which[0] == "gu" else \
        ops.gemv(W["qkv"], h, N=NQ, K=H, norm_w=W["ln"], eps=1e-5, y16=qkv, **k)
    res = {}
    for form in ("chain", "engine"):
        try:
            plan = ops.Plan()
            sync.zero_()
            for W in Ws:
                a = [mk(W, only_args=True)]
                if form == "chain":
                    ops.gemv_chain(a, sync[0], plan=plan)
                else:
                    ops.gemv_engine(a, sync[1], gran, timeout_ms=500, plan=plan)
            res[form] = "%.2f us (err word %d)" % (timeit(plan), int(sync[:, 1].sum()))
        except Exception as e:  # noqa: BLE001
            res[form] = "refused: " + str(e)[:100]
    print(f"  persistent forms, phase {which[0]}: chain {res['chain']} | engine {res['engine']}")
