"""GPU parity of the sampling head (usdm_sample_final) against oracle/sampling_oracle.py (itself pinned to HF's
logits warpers in tests/test_oracle_cpu.py): kept-id sets identical, probabilities to 1e-6, and the drawn id equal to
the oracle's for the same Philox (seed, step)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _state(dev, step=0):
    from usdm_amd import ops
    i32 = lambda n, v=0: torch.full((n,), v, dtype=torch.int32, device=dev)
    nxt, out, stp, pos = i32(1), i32(4096), i32(1, step), i32(1)
    return ops.decode_state(nxt, out, stp, pos), (nxt, out, stp, pos)


@pytest.mark.parametrize("V,T,k,p", [(42003, 1.0, 0, 1.0), (42003, 0.7, 50, 0.9), (42003, 1.3, 0, 0.95), (42003, 1.0, 200, 1.0),
                                     (517, 0.5, 10, 0.5), (1000, 2.0, 1000, 0.3)])
def test_sample_matches_oracle(dev, V, T, k, p):
    from usdm_amd import ops
    from oracle import sampling_oracle as so
    g = np.random.default_rng(V + k)
    x = (g.standard_normal(V) * 3).astype(np.float32)
    x = torch.from_numpy(x).to(torch.bfloat16).float().numpy()     # logits are bf16-valued on the path (ties do occur)
    x[g.integers(0, V, V // 10)] = -np.inf                          # banned ids arrive as -inf
    xd = torch.from_numpy(x).to(dev)
    probs = torch.zeros(V, device=dev)
    E = torch.randn(V, 64).to(torch.bfloat16).to(dev)
    h = torch.zeros(64, dtype=torch.bfloat16, device=dev)
    ref = so.filtered_probs(x, T, k, p)
    for step, seed in ((0, 0), (1, 0), (17, 12345), (255, 2 ** 40 + 7)):
        st, (nxt, out, stp, pos) = _state(dev, step)
        ops.sample_final(xd, st, temperature=T, top_k=k, top_p=p, seed=seed, probs_out=probs, embed=E, h_out=h, Hd=64)
        torch.cuda.synchronize()
        got = probs.cpu().numpy().astype(np.float64)
        assert ((got > 0) == (ref > 0)).all(), "kept-id set differs from the oracle"
        np.testing.assert_allclose(got, ref, rtol=3e-6, atol=1e-10)
        tok, _ = so.sample(x, step, T, k, p, seed)
        assert int(nxt.item()) == tok and int(out[step].item()) == tok
        assert int(stp.item()) == step + 1 and int(pos.item()) == 1
        assert torch.equal(h, E[tok])


def test_sample_distribution(dev):
    """4000 draws (step counter = draw index) from a 48-way distribution: chi-square against the filtered probabilities."""
    from usdm_amd import ops
    from oracle import sampling_oracle as so
    V, n = 48, 4000
    x = (np.random.default_rng(3).standard_normal(V) * 1.5).astype(np.float32)
    xd = torch.from_numpy(x).to(dev)
    st, (nxt, out, stp, pos) = _state(dev, 0)
    plan = ops.Plan()
    ops.sample_final(xd, st, temperature=0.9, top_k=0, top_p=0.97, seed=99, plan=plan)
    for _ in range(n):
        plan.run()
    torch.cuda.synchronize()
    toks = out[:n].cpu().numpy()
    assert int(stp.item()) == n
    p = so.filtered_probs(x, 0.9, 0, 0.97)
    assert (p[toks] > 0).all()
    cnt = np.bincount(toks, minlength=V).astype(np.float64)
    keep = p > 0
    chi2 = (((cnt - n * p) ** 2)[keep] / (n * p[keep])).sum()
    dof = keep.sum() - 1
    assert chi2 < dof + 5 * np.sqrt(2 * dof), (chi2, dof)
    # and the sequence is exactly the oracle's
    assert [so.sample(x, s, 0.9, 0, 0.97, 99)[0] for s in range(64)] == toks[:64].tolist()


def test_generate_sampling(dev):
    """generate(do_sample=True, ...) on a small random model: reproducible per seed, different across seeds, never a banned id,
    top_k=1 is the greedy path."""
    from usdm_amd.llm import USDMForCausalLM
    cfg = dict(vocab_size=1000, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768)
    m = USDMForCausalLM.random_init(cfg, dev, seed=5, ctx_max=256)
    m.reuse_prefix = False        # (every call below prefills the same 40 tokens: one plan per {greedy, sampling} is then the invariant)
    ids = torch.randint(0, 1000, (1, 40), generator=torch.Generator().manual_seed(1)).to(dev)
    bad = [[i] for i in range(0, 500)]
    kw = dict(input_ids=ids, max_new_tokens=24, do_sample=True, bad_words_ids=bad, temperature=1.5, top_p=0.95)
    a = m.generate(top_k=50, seed=1, **kw)
    b = m.generate(top_k=50, seed=1, **kw)
    c = m.generate(top_k=50, seed=2, **kw)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert (a[0, 40:] >= 500).all() and (c[0, 40:] >= 500).all()
    g1 = m.generate(top_k=1, **kw)
    g2 = m.generate(input_ids=ids, max_new_tokens=24, do_sample=False, bad_words_ids=bad)
    assert torch.equal(g1, g2)
    # one captured decode graph / prefill plan serves every (temperature, top-k, top-p, seed): the knobs live in device memory
    assert len(m._decodes) == 2 and sum(1 for k in m._prefill_plans if k[2]) == 1
    # default seed: drawn from torch's global generator -> calls differ, reproducible under manual_seed (HF sampling behaviour)
    torch.manual_seed(1234); d1 = m.generate(top_k=50, **kw)
    d2 = m.generate(top_k=50, **kw)
    torch.manual_seed(1234); d3 = m.generate(top_k=50, **kw)
    assert torch.equal(d1, d3) and not torch.equal(d1, d2)


def test_sample_all_banned_or_nan_never_leaves_the_vocabulary(dev):
    """with zero kept mass (every logit -inf, or NaN) the kernel used to emit id_offset-1 and read the embedding
    table out of range.  Now: arg-max of the finite logits, else id 0."""
    from usdm_amd import ops
    V = 777
    E = torch.randn(V, 64, device=dev).to(torch.bfloat16)
    h = torch.zeros(64, dtype=torch.bfloat16, device=dev)
    nxt, outt, step, pos = (torch.zeros(n, dtype=torch.int32, device=dev) for n in (1, 8, 1, 1))
    for fill, expect in ((float("-inf"), 0), (float("nan"), 0)):
        x = torch.full((V,), fill, device=dev)
        st = ops.decode_state(nxt, outt, step, pos)
        step.zero_()
        ops.sample_final(x, st, temperature=1.0, top_k=0, top_p=0.9, seed=3, embed=E, h_out=h, Hd=64)
        torch.cuda.synchronize()
        assert int(nxt.item()) == expect and torch.equal(h, E[expect])
    x = torch.full((V,), float("nan"), device=dev)
    x[123] = -5.0                                   # a single finite logit among NaNs: mass is NaN -> falls back to the arg-max
    st = ops.decode_state(nxt, outt, step, pos)
    ops.sample_final(x, st, temperature=1.0, top_k=0, top_p=1.0, seed=3, embed=E, h_out=h, Hd=64)
    torch.cuda.synchronize()
    assert 0 <= int(nxt.item()) < V
