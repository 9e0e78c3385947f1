"""Batched decode.  VALU form (<= 4 sequences): usdm_gemv_batch must reproduce usdm_gemv bit for bit per item, and
generate_batch(group=4) must return exactly what generate() returns for each prompt on its own.  Matrix-core form (<= 16 sequences,
round 4): the same rounding points with K summed in another order - per item within one bf16 ulp of usdm_gemv and of float64, and
generate_batch against the CPU oracle under its near-tie rule at B = 8 and 16."""
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
    batch = m.generate_batch(prompts, max_new_tokens=30, bad_words_ids=bad, eos_token_id=eos, group=4)      # VALU form: bit-identical
    assert len(batch) == len(prompts)
    for s, b in zip(singles, batch):
        assert torch.equal(s, b), (s.shape, b.shape)
    assert singles[1].shape[1] <= 17 + 10


def test_generate_batch_falls_back_to_four_when_k_does_not_fit_the_matrix_core_form(dev):
    """The matrix-core form splits K over 8 waves in chunks of 32: a model whose projections have a K that is not a multiple of 256
    (hidden 384 here) must not fail on a list of more than 4 prompts - max_batch() says 4 and the list runs in groups of 4 on the
    VALU kernel, every sequence bit-identical with generate()."""
    from usdm_amd.llm import USDMForCausalLM
    cfg = dict(vocab_size=1000, hidden_size=384, intermediate_size=640, num_hidden_layers=2, num_attention_heads=3,
               num_key_value_heads=3, head_dim=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768)
    m = USDMForCausalLM.random_init(cfg, dev, seed=9, ctx_max=128)
    assert m.max_batch() == 4
    gen = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, 1000, (1, L), generator=gen).to(dev) for L in (21, 9, 33, 17, 26, 12)]
    singles = [m.generate(input_ids=p, max_new_tokens=12) for p in prompts]
    batch = m.generate_batch(prompts, max_new_tokens=12)
    for s_, b_ in zip(singles, batch):
        assert torch.equal(s_, b_)


@pytest.mark.parametrize("group", [4, 16])
def test_generate_batch_vs_oracle(dev, group):
    """Batched decode against the CPU ORACLE (not against the single-sequence HIP path): every sequence of a 5-prompt batch
    (groups of 4 + 1 on the VALU kernel / one group of 5 on the matrix cores; ragged lengths, ban mask, an EOS that stops one
    sequence early) equals oracle greedy generation of that prompt, up to oracle near-ties."""
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
    batch = m.generate_batch([p[None].to(dev) for p in prompts], max_new_tokens=30, bad_words_ids=bad, eos_token_id=eos, group=group)
    firsts = []
    for p, (ref, ref_logits), out in zip(prompts, refs, batch):
        firsts.append(check_against_oracle(out[0].tolist(), ref, ref_logits, p.numel()))
    print("batched decode vs oracle: first differences (None = identical):", firsts)
    assert len(refs[1][0]) <= 17 + 10


def _close_bf16(y, ref, what, ulps=1, mag=None):
    """within `ulps` bf16 ulps of the larger magnitude (+ the f32 accumulation difference near zero); mag: magnitude of a term that
    was added after a rounding (the residual: an ulp of the rounded product can be many ulps of a sum that cancels)"""
    yf, rf = y.float(), ref.float()
    m = torch.maximum(yf.abs(), rf.abs())
    if mag is not None:
        m = torch.maximum(m, mag.float().abs().to(m.device))
    tol = ulps * 2.0 ** -7 * m + 2e-5 * rf.abs().max()
    bad = (yf - rf).abs() > tol
    assert not bool(bad.any()), f"{what}: {int(bad.sum())} outputs beyond one bf16 ulp, max diff {(yf - rf).abs().max().item()}"


@pytest.mark.parametrize("N,K,act,norm,res", [(6144, 4096, 0, True, False), (4096, 4096, 0, False, True), (28672, 4096, 3, True, False),
                                              (4096, 14336, 0, False, True), (100, 512, 0, False, False), (96, 1792, 3, False, False),
                                              (1000, 4096, 0, True, True), (4096, 3584, 0, False, False)])
@pytest.mark.parametrize("nb", [1, 5, 8, 16])
def test_gemv_batch_matrix_core_form(dev, N, K, act, norm, res, nb):
    """usdm_gemv_batch form 1 (v_mfma_f32_16x16x32_bf16, weights as the A operand) against usdm_gemv per item and against float64:
    the 7B's four projection shapes (12-row tiles for N = 6144, the 8 + 8 SwiGLU tiles, the streamed-activation variant for
    K = 14336), ragged N, a K that does not fill the 16 waves evenly, K slices longer than the held 8 chunks."""
    from usdm_amd import ops
    bf = torch.bfloat16
    W = _r((N, K), 1, K ** -0.5).to(bf).to(dev)
    X = _r((nb, K), 2).to(bf).to(dev)
    g = (1 + 0.1 * _r((K,), 4)).to(bf).float().to(dev) if norm else None
    nout = N // 2 if act == 3 else N
    R = _r((nb, nout), 3).to(bf).to(dev) if res else None
    Yb = torch.full((nb, nout), float("nan"), dtype=bf, device=dev)
    ops.gemv_batch(W, X, nb=nb, N=N, K=K, x_bs=K, y_bs=nout, res_bs=nout, norm_w=g, act=act, residual=R, y16=Yb, form=1)
    assert bool(torch.isfinite(Yb.float()).all()), "outputs left unwritten"
    for b in range(nb):
        y = torch.zeros(nout, dtype=bf, device=dev)
        ops.gemv(W, X[b], N=N, K=K, norm_w=g, act=act, residual=R[b] if res else None, y16=y)
        _close_bf16(Yb[b], y, f"item {b} vs usdm_gemv", ulps=3 if act == 3 else (2 if res else 1), mag=R[b] if res else None)   # SwiGLU: three chained roundings
    # float64 with the same rounding points
    x = X.double().cpu()
    if norm:
        xn = (x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-5)).to(bf).double()
        x = (xn * g.double().cpu()).to(bf).double()
    acc = x @ W.double().cpu().T
    if act == 3:
        gt, up = acc.reshape(nb, -1, 2, 16)[:, :, 0].reshape(nb, -1).to(bf).double(), acc.reshape(nb, -1, 2, 16)[:, :, 1].reshape(nb, -1).to(bf).double()
        ref = ((gt * torch.sigmoid(gt)).to(bf).double() * up).to(bf)
    else:
        ref = acc.to(bf)
        if res:
            ref = (ref.double() + R.double().cpu()).to(bf)
    _close_bf16(Yb.cpu(), ref, "vs float64", ulps=3 if act == 3 else (2 if res else 1), mag=R.cpu() if res else None)


@pytest.mark.parametrize("N,K,res", [(4096, 14336, True), (4096, 6144, False), (1000, 16384, True), (200, 14336, False)])
@pytest.mark.parametrize("nb", [5, 16])
def test_gemv_batch_k_split_over_workgroups(dev, N, K, res, nb):
    """Matrix-core form with K split over workgroups (usdm_gemv_batch ks_part / ks_cnt; down_proj of the 7B: K = 14336 = 7 slices of
    2048): partial tiles meet in device memory, the last arriver sums them in slice order.  Against float64 with the same rounding
    points; three launches in a row on the same scratch (the counters must come back to zero, results must be REPRODUCIBLE bit for
    bit whichever workgroup arrives last), in-place residual (y16 = residual) as the decode step uses it."""
    from usdm_amd import ops
    bf = torch.bfloat16
    W = _r((N, K), 1, K ** -0.5).to(bf).to(dev)
    X = _r((nb, K), 2).to(bf).to(dev)
    R = _r((nb, N), 3).to(bf).to(dev) if res else None
    ksf = ops.gemv_batch_ks_floats(N, K)
    assert ksf == -(-N // 16) * (K // 2048) * 256
    part = torch.full((ksf,), float("nan"), device=dev)
    cnt = torch.zeros(-(-N // 16), dtype=torch.int32, device=dev)
    outs = []
    for rep in range(3):
        Y = R.clone() if res else torch.full((nb, N), float("nan"), dtype=bf, device=dev)
        ops.gemv_batch(W, X, nb=nb, N=N, K=K, x_bs=K, y_bs=N, res_bs=N, residual=Y if res else None, y16=Y, form=1, ks=(part, cnt))
        assert int(cnt.abs().sum()) == 0, "tile counters not back at zero"
        outs.append(Y.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), "the K-split sum depends on the arrival order"
    assert bool(torch.isfinite(part).all())
    ref = (X.double().cpu() @ W.double().cpu().T).to(bf)
    if res:
        ref = (ref.double() + R.double().cpu()).to(bf)
    _close_bf16(outs[0].cpu(), ref, "K-split vs float64", ulps=2 if res else 1, mag=R.cpu() if res else None)
    # and against the unsplit kernel (form 5): same rounding points, another summation order
    Y5 = R.clone() if res else torch.zeros(nb, N, dtype=bf, device=dev)
    ops.gemv_batch(W, X, nb=nb, N=N, K=K, x_bs=K, y_bs=N, res_bs=N, residual=Y5 if res else None, y16=Y5, form=5, ks=(part, cnt))
    _close_bf16(outs[0], Y5, "K-split vs unsplit", ulps=2 if res else 1, mag=R if res else None)


@pytest.mark.parametrize("nb", [3, 16])
def test_gemv_batch_matrix_core_lm_head(dev, nb):
    """lm_head mode of the matrix-core form: bf16 logits, ban mask (whole banned tiles are not streamed), one arg-max partial per
    workgroup, ties -> lowest id; the token usdm_argmax_final picks equals the arg-max of the masked logits the launch wrote."""
    from usdm_amd import ops
    bf = torch.bfloat16
    V, K = 42003, 4096
    W, X = _r((V, K), 5, K ** -0.5).to(bf).to(dev), _r((nb, K), 6).to(bf).to(dev)
    g = (1 + 0.1 * _r((K,), 4)).to(bf).float().to(dev)
    ban = torch.zeros(V, dtype=torch.uint8, device=dev)
    ban[:32002] = 1; ban[28705] = 0; ban[40000::7] = 1          # the text -> unit round's mask shape + scattered bans
    n = ops.gemv_nblocks(V)
    pv, pi = torch.zeros(nb, n, device=dev), torch.zeros(nb, n, dtype=torch.int32, device=dev)
    lg = torch.full((nb, V), float("nan"), device=dev)
    ops.gemv_batch(W, X, nb=nb, N=V, K=K, x_bs=K, y_bs=V, part_bs=n, norm_w=g, ban=ban, part_val=pv, part_idx=pi, y32=lg, form=1)
    banned = ban.bool().cpu()
    assert bool(torch.isinf(lg.cpu()[:, banned]).all()) and bool(torch.isfinite(lg.cpu()[:, ~banned]).all())
    for b in range(nb):
        l1 = torch.zeros(V, device=dev)
        pv1, pi1 = torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
        ops.gemv(W, X[b], N=V, K=K, norm_w=g, ban=ban, part_val=pv1, part_idx=pi1, y32=l1)
        ok = ~banned
        _close_bf16(lg[b].cpu()[ok].to(bf), l1.cpu()[ok].to(bf), f"logits of item {b}")
        best = int(pi[b][pv[b] == pv[b].max()].min().item())       # usdm_argmax_final's rule: ties between partials -> lowest id
        row = lg[b].cpu()
        assert row[best] == row.max() and best == int((row == row.max()).nonzero()[0]), "partials do not hold the lowest-id arg-max of the logits written"


@pytest.mark.parametrize("B", [8, 16])
def test_generate_batch_matrix_cores_vs_oracle(dev, B):
    """B = 8 / 16 sequences per decode step (one group on the matrix-core form) against the CPU oracle under the near-tie rule."""
    from oracle import mistral_oracle as MO
    from tests._greedy_compare import check_against_oracle
    from usdm_amd.llm import USDMForCausalLM
    cfg = dict(vocab_size=1000, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768)
    sd = MO.random_state_dict(cfg, seed=19)
    m = USDMForCausalLM.from_state_dict(sd, cfg, dev, ctx_max=256)
    gen = torch.Generator().manual_seed(5)
    prompts = [torch.randint(0, 1000, (int(L),), generator=gen) for L in torch.randint(12, 70, (B,), generator=gen)]
    bad = [[i] for i in range(0, 300)]
    refs = [MO.greedy_generate(sd, cfg, p, 24, bad_words_ids=bad, return_logits=True) for p in prompts]
    batch = m.generate_batch([p[None].to(dev) for p in prompts], max_new_tokens=24, bad_words_ids=bad)
    firsts = [check_against_oracle(out[0].tolist(), ref, ref_logits, p.numel()) for p, (ref, ref_logits), out in zip(prompts, refs, batch)]
    print(f"B = {B} on the matrix cores vs oracle: first differences (None = identical):", firsts)
