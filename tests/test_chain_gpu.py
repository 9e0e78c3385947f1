"""GPU: usdm_gemv_chain (consecutive decode projections in one persistent launch, weight stream running across the phase
boundaries) against the same projections as separate usdm_gemv launches: BIT-IDENTICAL outputs at the 7B shapes, under graph
replay (monotonic barrier counters), and token-identical generation."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
ACT_SWIGLU = 3


def _setup(dev):
    g = torch.Generator(device=dev).manual_seed(7)
    bf = torch.bfloat16
    H, I, NQ = 4096, 14336, 6144
    r = lambda *s, sc: (torch.randn(*s, device=dev, generator=g) * sc).to(bf)
    from usdm_amd.llm import _pack_gate_up
    W = dict(o=r(H, H, sc=H ** -0.5), gu=_pack_gate_up(r(I, H, sc=H ** -0.5), r(I, H, sc=H ** -0.5)), down=r(H, I, sc=I ** -0.5),
             qkv=r(NQ, H, sc=H ** -0.5), ln1=1 + 0.1 * torch.randn(H, device=dev, generator=g), ln2=1 + 0.1 * torch.randn(H, device=dev, generator=g))
    x = dict(h=r(H, sc=1.0), ao=r(H, sc=1.0))
    return W, x, (H, I, NQ)


def _phases(ops, W, h, ao, act, qkv, dims, n, **kw):
    H, I, NQ = dims
    ph = [lambda **k: ops.gemv(W["o"], ao, N=H, K=H, residual=h, y16=h, **k),
          lambda **k: ops.gemv(W["gu"], h, N=2 * I, K=H, norm_w=W["ln2"], eps=1e-5, act=ACT_SWIGLU, y16=act, **k),
          lambda **k: ops.gemv(W["down"], act, N=H, K=I, residual=h, y16=h, **k),
          lambda **k: ops.gemv(W["qkv"], h, N=NQ, K=H, norm_w=W["ln1"], eps=1e-5, y16=qkv, **k)]
    return ph[:n]


@pytest.mark.parametrize("nph", [3, 4])
def test_chain_bit_identical_to_separate_launches(dev, nph):
    from usdm_amd import ops
    from usdm_amd.graph import GraphedPlan
    W, x, dims = _setup(dev)
    H, I, NQ = dims
    bf = torch.bfloat16
    mk = lambda: (x["h"].clone(), x["ao"].clone(), torch.zeros(I, dtype=bf, device=dev), torch.zeros(NQ, dtype=bf, device=dev))
    # reference: one launch per projection
    h, ao, act, qkv = mk()
    for f in _phases(ops, W, h, ao, act, qkv, dims, nph):
        f()
    ref = (h.clone(), act.clone(), qkv.clone())
    # chain, eager
    sync = torch.zeros(8, dtype=torch.int32, device=dev)
    h2, ao2, act2, qkv2 = mk()
    plan = ops.Plan()
    ops.gemv_chain([f(only_args=True) for f in _phases(ops, W, h2, ao2, act2, qkv2, dims, nph)], sync, plan=plan)
    plan.run()
    torch.cuda.synchronize()
    assert sync.tolist()[:2] == [1, 0], sync.tolist()                       # generation advanced, no timeout
    assert torch.equal(h2, ref[0]) and torch.equal(act2, ref[1]) and (nph < 4 or torch.equal(qkv2, ref[2]))
    # replayed as a hipGraph, several times, from the same inputs: the barrier counters are monotonic, nothing is reset
    gp = GraphedPlan(plan)
    for rep in range(5):
        h2.copy_(x["h"]); act2.zero_(); qkv2.zero_()
        gp.run()
        torch.cuda.synchronize()
        assert torch.equal(h2, ref[0]) and torch.equal(act2, ref[1]) and (nph < 4 or torch.equal(qkv2, ref[2])), rep
    s = sync.tolist()
    assert s[0] == 6 and s[1] == 0 and s[2] == 6 * 512 and s[3] == 6 * 512, s
    # timing, informational
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    sep = ops.Plan()
    for f in _phases(ops, W, h, ao, act, qkv, dims, nph):
        f(plan=sep)
    gs = GraphedPlan(sep)
    for g_ in (gs, gp):
        for _ in range(3):
            g_.run()
    res = []
    for g_ in (gs, gp):
        e0.record()
        for _ in range(20):
            g_.run()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"{nph} phases (hot weights, same matrices every run): separate launches {res[0]:.1f} us, chain {res[1]:.1f} us")


def test_generation_with_chained_decode_is_token_identical(dev):
    """4-layer model at the 7B widths: the chained decode step (o -> gate/up -> down [-> next qkv]) must reproduce the launch-per-
    projection step token for token (same per-row arithmetic)."""
    from oracle import mistral_oracle as MO
    from usdm_amd import synth
    from usdm_amd.llm import USDMForCausalLM
    cfg = dict(MO.MISTRAL_7B_USDM, num_hidden_layers=4)
    sd = synth.random_llm_state_dict(cfg, dev, seed=13)
    ids = torch.randint(32002, 42002, (1, 50), generator=torch.Generator().manual_seed(2)).to(dev)
    outs = {}
    for mode in ("0", "3", "4"):
        os.environ["USDM_GEMV_CHAIN"] = mode
        try:
            m = USDMForCausalLM.from_state_dict(sd, cfg, dev, ctx_max=128)
        finally:
            os.environ.pop("USDM_GEMV_CHAIN")
        assert m.chain == int(mode)
        outs[mode] = m.generate(input_ids=ids, max_new_tokens=24)[0].tolist()
        if mode != "0":
            assert any(w == "usdm_gemv_chain" for w, _, _ in m._decode.plan.calls)
            assert int(m.chain_sync[:, 1].sum().item()) == 0
    assert outs["3"] == outs["0"] and outs["4"] == outs["0"]


@pytest.mark.parametrize("nph", [1, 2, 3, 4])
def test_engine_bit_identical_to_separate_launches(dev, nph):
    """usdm_gemv_engine (LDS-DMA loader + consumer waves per CU, granule hand-offs between phases) vs one usdm_gemv launch per
    projection at the 7B shapes: bit-identical vectors, also when replayed as a hipGraph (generation-tagged granules)."""
    from usdm_amd import ops
    from usdm_amd.graph import GraphedPlan
    W, x, dims = _setup(dev)
    H, I, NQ = dims
    bf = torch.bfloat16
    mk = lambda: (x["h"].clone(), x["ao"].clone(), torch.zeros(I, dtype=bf, device=dev), torch.zeros(NQ, dtype=bf, device=dev))
    h, ao, act, qkv = mk()
    for f in _phases(ops, W, h, ao, act, qkv, dims, nph):
        f()
    ref = (h.clone(), act.clone(), qkv.clone())
    sync = torch.zeros(8, dtype=torch.int32, device=dev)
    gran = torch.zeros(3 * 8192, dtype=torch.int64, device=dev)
    h2, ao2, act2, qkv2 = mk()
    plan = ops.Plan()
    ops.gemv_engine([f(only_args=True) for f in _phases(ops, W, h2, ao2, act2, qkv2, dims, nph)], sync, gran, timeout_ms=500, plan=plan)

    def check(tag):
        torch.cuda.synchronize()
        assert int(sync[1].item()) == 0, f"{tag}: engine wait timed out"
        assert torch.equal(h2, ref[0]), f"{tag}: residual stream differs: {(h2 != ref[0]).sum().item()} of {H}"
        if nph >= 2:
            assert torch.equal(act2, ref[1]), f"{tag}: SwiGLU output differs: {(act2 != ref[1]).sum().item()} of {I}"
        if nph >= 4:
            assert torch.equal(qkv2, ref[2]), f"{tag}: qkv differs: {(qkv2 != ref[2]).sum().item()} of {NQ}"
    plan.run()
    check("eager")
    gp = GraphedPlan(plan)
    for rep in range(4):
        h2.copy_(x["h"]); act2.zero_(); qkv2.zero_()
        gp.run()
        check(f"replay {rep}")
    assert int(sync[0].item()) == (5 if nph > 1 else 0)


def test_generation_with_engine_decode_is_token_identical(dev):
    from oracle import mistral_oracle as MO
    from usdm_amd import synth
    from usdm_amd.llm import USDMForCausalLM
    cfg = dict(MO.MISTRAL_7B_USDM, num_hidden_layers=4)
    sd = synth.random_llm_state_dict(cfg, dev, seed=13)
    ids = torch.randint(32002, 42002, (1, 50), generator=torch.Generator().manual_seed(2)).to(dev)
    outs = {}
    for mode in ("0", "e3", "e4"):
        os.environ["USDM_GEMV_CHAIN"] = mode
        try:
            m = USDMForCausalLM.from_state_dict(sd, cfg, dev, ctx_max=128)
        finally:
            os.environ.pop("USDM_GEMV_CHAIN")
        outs[mode] = m.generate(input_ids=ids, max_new_tokens=24)[0].tolist()
        if mode != "0":
            assert any(w == "usdm_gemv_engine" for w, _, _ in m._decode.plan.calls)
            assert int(m.chain_sync[:, 1].sum().item()) == 0
    assert outs["e3"] == outs["0"] and outs["e4"] == outs["0"]
