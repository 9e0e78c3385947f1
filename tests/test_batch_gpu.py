"""Batched decode: usdm_gemv_batch must reproduce usdm_gemv bit for bit per item, and generate_batch must
return exactly what generate() returns for each prompt on its own."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _r(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("N,K,act,norm,res", [(6144, 4096, 0, True, False), (4096, 4096, 0, False, True), (1024, 512, 3, True, False),
                                              (4096, 14336, 0, False, True), (100, 512, 0, False, False), (96, 1792, 3, False, False)])
@pytest.mark.parametrize("nb", [1, 2, 3, 4])
def test_gemv_batch_equals_gemv(dev, N, K, act, norm, res, nb):
    from usdm_amd import ops
    bf = torch.bfloat16
    W = _r((N, K), 1, K ** -0.5).to(bf).to(dev)
    X = _r((nb, K), 2).to(bf).to(dev)
    g = (1 + 0.1 * _r((K,), 4)).to(dev) if norm else None
    nout = N // 2 if act == 3 else N
    R = _r((nb, nout), 3).to(bf).to(dev) if res else None
    Yb = torch.zeros(nb, nout, dtype=bf, device=dev)
    ops.gemv_batch(W, X, nb=nb, N=N, K=K, x_bs=K, y_bs=nout, res_bs=nout, norm_w=g, act=act, residual=R, y16=Yb)
    for b in range(nb):
        y = torch.zeros(nout, dtype=bf, device=dev)
        ops.gemv(W, X[b], N=N, K=K, norm_w=g, act=act, residual=R[b] if res else None, y16=y)
        assert torch.equal(y, Yb[b]), (b, (y.float() - Yb[b].float()).abs().max())


def test_gemv_batch_lm_head(dev):
    from usdm_amd import ops
    bf = torch.bfloat16
    V, K, nb = 1003, 512, 3
    W, X = _r((V, K), 5, K ** -0.5).to(bf).to(dev), _r((nb, K), 6).to(bf).to(dev)
    ban = torch.zeros(V, dtype=torch.uint8, device=dev); ban[::7] = 1
    n = ops.gemv_nblocks(V)
    pv, pi = torch.zeros(nb, n, device=dev), torch.zeros(nb, n, dtype=torch.int32, device=dev)
    ops.gemv_batch(W, X, nb=nb, N=V, K=K, x_bs=K, part_bs=n, ban=ban, part_val=pv, part_idx=pi)
    for b in range(nb):
        pv1, pi1 = torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
        ops.gemv(W, X[b], N=V, K=K, ban=ban, part_val=pv1, part_idx=pi1)
        assert torch.equal(pv1, pv[b]) and torch.equal(pi1, pi[b])


def test_generate_batch_equals_generate(dev):
    from usdm_amd.llm import USDMForCausalLM
    cfg = dict(vocab_size=1000, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768)
    m = USDMForCausalLM.random_init(cfg, dev, seed=7, ctx_max=256)
    gen = torch.Generator().manual_seed(2)
    prompts = [torch.randint(0, 1000, (1, L), generator=gen).to(dev) for L in (40, 17, 65, 33, 50)]   # 5 prompts -> groups of 4 + 1
    bad = [[i] for i in range(0, 300)]
    singles = [m.generate(input_ids=p, max_new_tokens=30, bad_words_ids=bad) for p in prompts]
    eos = int(singles[1][0, 17 + 9])                      # make prompt 1 stop early: its 10th generated token is the EOS id
    singles = [m.generate(input_ids=p, max_new_tokens=30, bad_words_ids=bad, eos_token_id=eos) for p in prompts]
    batch = m.generate_batch(prompts, max_new_tokens=30, bad_words_ids=bad, eos_token_id=eos)
    assert len(batch) == len(prompts)
    for s, b in zip(singles, batch):
        assert torch.equal(s, b), (s.shape, b.shape)
    assert singles[1].shape[1] <= 17 + 10


def test_generate_batch_vs_oracle(dev):
    """Batched decode against the CPU ORACLE (not against the single-sequence HIP path): every sequence of a 5-prompt batch
    (groups of 4 + 1, ragged lengths, ban mask, an EOS that stops one sequence early) equals oracle greedy generation of that
    prompt, up to oracle near-ties."""
    from oracle import mistral_oracle as MO
    from tests._greedy_compare import check_against_oracle
    from usdm_amd.llm import USDMForCausalLM
    cfg = dict(vocab_size=1000, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768)
    sd = MO.random_state_dict(cfg, seed=17)
    m = USDMForCausalLM.from_state_dict(sd, cfg, dev, ctx_max=256)
    gen = torch.Generator().manual_seed(4)
    prompts = [torch.randint(0, 1000, (L,), generator=gen) for L in (40, 17, 65, 33, 50)]
    bad = [[i] for i in range(0, 300)]
    probe = MO.greedy_generate(sd, cfg, prompts[1], 30, bad_words_ids=bad)
    eos = probe[17 + 9]                                        # prompt 1 stops at its 10th generated token (or earlier)
    refs = [MO.greedy_generate(sd, cfg, p, 30, bad_words_ids=bad, eos_token_id=eos, return_logits=True) for p in prompts]
    batch = m.generate_batch([p[None].to(dev) for p in prompts], max_new_tokens=30, bad_words_ids=bad, eos_token_id=eos)
    firsts = []
    for p, (ref, ref_logits), out in zip(prompts, refs, batch):
        firsts.append(check_against_oracle(out[0].tolist(), ref, ref_logits, p.numel()))
    print("batched decode vs oracle: first differences (None = identical):", firsts)
    assert len(refs[1][0]) <= 17 + 10
