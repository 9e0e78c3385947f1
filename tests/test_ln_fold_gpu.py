"""The folded LayerNorm (usdm_gemm stats_out / ln_mode; reference block: src/decoder/voicebox/model/networks.py:236-266) under
adversarial row statistics: a DC offset |mean| / sigma in {0, 1, 10, 100} and one channel at 100 sigma.

What must hold:
  * the row statistics (per-tile sum + M2, merged pairwise) are accurate at ANY offset: the f32 residual form (ln_mode 2) stays at
    f32 accuracy against float64;
  * the GELU consumer (ln_mode 1) multiplies rows rounded to bf16 BEFORE centring: inside the guard ratio it must hold the same
    tolerance as the unfolded path, beyond it the guard word must be raised (never a silent loss of accuracy);
  * on the real plan path (R >= 2048, so the fold is live) a model whose out-proj bias carries a large DC offset trips the guard,
    falls back to the LayerNorm kernel by itself and meets the 1e-2 per-evaluation tolerance against the fp32 oracle.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _rows(M, H, ratio, outlier, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, H, generator=g) * 1.7          # sigma 1.7
    if outlier:
        x[:, 37] += 100 * 1.7                          # one "massive activation" channel at 100 sigma
    return x + ratio * 1.7                             # DC offset = ratio sigma


@pytest.mark.parametrize("ratio,outlier", [(0.0, False), (1.0, False), (0.0, True), (10.0, False), (100.0, False)])
def test_folded_layernorm_adversarial_rows(dev, ratio, outlier):
    from usdm_amd import ops
    from usdm_amd._lib import ACT_GELU
    M, H, I = 2236, 1024, 1024
    nt = H // 128
    x = _rows(M, H, ratio, outlier, 5)
    # producer: x1 = 0 * Wo + 0 + res = the adversarial rows, through the real out-proj epilogue (statistics + bf16 copy)
    A, Wo = torch.zeros(M, H, dtype=bf, device=dev), torch.zeros(H, H, dtype=bf, device=dev)
    x32, x16 = torch.zeros(M, H, device=dev), torch.zeros(M, H, device=dev, dtype=bf)
    st = torch.full((M, nt, 2), float("nan"), device=dev)
    ops.gemm(A, Wo, M=M, N=H, Kc=H, residual=x.to(dev), ldr=H, out32=x32, out16=x16, stats_out=st)
    assert torch.equal(x32.cpu(), x)
    g = torch.Generator().manual_seed(6)
    gam, bet = 1 + 0.2 * torch.randn(H, generator=g), 0.3 * torch.randn(H, generator=g)
    W1, b1 = torch.randn(I, H, generator=g) * 0.05, torch.randn(I, generator=g)
    ln64 = torch.nn.functional.layer_norm(x.double(), (H,), gam.double(), bet.double(), 1e-5)
    guard = torch.zeros(1, dtype=torch.int32, device=dev)
    lnk = dict(stats=st, nt=nt, C=H, eps=1e-5, guard=guard, guard_ratio=2.0)
    # ---- f32 residual consumer: accurate at every offset (this is what sum(x^2) - mean^2 could not do)
    F_, W2, b2 = torch.zeros(M, I, dtype=bf, device=dev), torch.zeros(H, I, dtype=bf, device=dev), torch.zeros(H, device=dev)
    y = torch.zeros(M, H, device=dev)
    ops.gemm(F_, W2, M=M, N=H, Kc=I, bias=b2, residual=x32, ldr=H, out32=y, ln=dict(mode=2, gamma=gam.to(dev), beta=bet.to(dev), **lnk))
    err2 = (y.double().cpu() - ln64).abs().max().item() / ln64.abs().max().item()
    # ---- GELU consumer on the pre-rounded rows, and the unfolded path (LayerNorm kernel -> bf16 -> GEMM) beside it
    w1g = (W1 * gam[None]).to(bf).to(dev).contiguous()
    c1, d1 = w1g.float().sum(1).contiguous(), (b1 + W1 @ bet).to(dev).contiguous()
    guard.zero_()
    f_fold = torch.zeros(M, I, device=dev, dtype=bf)
    ops.gemm(x16, w1g, M=M, N=I, Kc=H, bias=d1, act=ACT_GELU, out16=f_fold, ln=dict(mode=1, c=c1, **lnk))
    tripped = int(guard.item())
    h16 = torch.zeros(M, H, device=dev, dtype=bf)
    ops.norm(x32, gam.to(dev), bet.to(dev), rows=M, C=H, out16=h16)
    f_plain = torch.zeros(M, I, device=dev, dtype=bf)
    ops.gemm(h16, W1.to(bf).to(dev).contiguous(), M=M, N=I, Kc=H, bias=b1.to(dev), act=ACT_GELU, out16=f_plain)
    ref = torch.nn.functional.gelu(ln64 @ W1.double().T + b1.double())
    rel = lambda o: ((o.double().cpu() - ref).norm() / ref.norm()).item()
    e_fold, e_plain = rel(f_fold), rel(f_plain)
    print(f"|mean|/sigma {ratio:5.1f} outlier {outlier}: ln_mode 2 max rel err {err2:.2e}; GELU consumer rel L2 folded {e_fold:.2e} / "
          f"LayerNorm kernel {e_plain:.2e}; guard {'RAISED' if tripped else 'silent'}")
    assert err2 <= 2e-5, "row statistics lost accuracy under a DC offset"
    if ratio <= 2.0:
        assert not tripped, "guard raised inside its bound"
        assert e_fold <= max(2.5 * e_plain, 4e-3), "folded GELU consumer outside the unfolded path's tolerance"
    else:
        assert tripped, "rows beyond the guard ratio were accepted silently"


def _model(dev, layers, bias_dc, outlier):
    from oracle import voicebox_oracle as VO
    from usdm_amd.voicebox.model import Voicebox
    cfg = dict(VO.VOICEBOX_CFG, num_hidden_layers=layers)
    sd = VO.random_state_dict(cfg, 3)
    for l in range(layers):
        b = sd[f"estimator.layers.{l}.attention.out_proj.bias"]
        b += bias_dc                       # a DC offset in x1 = h + attn Wo + bo that LayerNorm 1 has to remove
        if outlier:
            b[5] += 150.0
    kw = {k: cfg[k] for k in cfg if k != "sigma_min"}
    m = Voicebox(**kw, attention_dropout=0.0, activation_dropout=0.1, hidden_dropout=0.0, solver="euler", sigma_min=cfg["sigma_min"])
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval(), sd, cfg


@pytest.mark.parametrize("bias_dc,outlier,expect_fallback", [(0.0, False, False), (1.0, True, False), (40.0, False, True)])
def test_real_plan_fold_guard_and_fallback(dev, bias_dc, outlier, expect_fallback):
    """Two full-width layers at B = 2, S = 1117 (R = 2236: the folded plan).  A benign model keeps the fold; a model whose residual
    stream carries a DC offset of ~20 sigma trips the guard, and the evaluation that returns is the fallback's."""
    from oracle import voicebox_oracle as VO
    m, sd, cfg = _model(dev, 2, bias_dc, outlier)
    S = 1117
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, cfg["n_tokens"], (2, S), generator=g)
    y, cond = torch.randn(2, 80, S, generator=g), torch.randn(2, 80, S, generator=g)
    t, lens = torch.tensor([0.3, 0.7]).reshape(2, 1, 1), torch.tensor([S, S])
    ref = VO.estimator_forward(sd, cfg, x, y, cond, t, lens)
    est = m.estimator
    assert est.ln_fold_ok
    out = est(x.to(dev), y.to(dev), cond.to(dev), t.to(dev), lens.to(dev))
    rel = ((out.double().cpu() - ref.double()).norm() / ref.double().norm()).item()
    print(f"out-proj bias DC {bias_dc}, outlier {outlier}: fold {'kept' if est.ln_fold_ok else 'switched off by the guard'}, rel L2 {rel:.2e}")
    assert est.ln_fold_ok == (not expect_fallback)
    assert rel <= 1e-2
    if expect_fallback:      # what the guard protected against: the same input through the folded plan, guard ignored
        est.ln_fold_ok, est.ln_guard_ratio = True, 1e30
        est._plans.clear()
        bad = est(x.to(dev), y.to(dev), cond.to(dev), t.to(dev), lens.to(dev))
        rb = ((bad.double().cpu() - ref.double()).norm() / ref.double().norm()).item()
        print(f"   the folded plan on this input (guard disabled): rel L2 {rb:.2e}")
