"""GPU: plan caches are bounded and bucketed (VERDICT r01 #6 / ADVICE r01): two lengths inside ONE bucket both match the CPU
oracle, running them alternately on the shared workspace changes nothing, and 50 random lengths leave the caches bounded."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(out, ref):
    return ((out.double().cpu() - ref.double()).norm() / ref.double().norm()).item()


def test_voicebox_two_lengths_in_one_bucket_match_oracle_and_cache_is_bounded(dev):
    from oracle import voicebox_oracle as VO
    from tests.golden.configs import SMALL_VB
    from tests.test_voicebox_gpu import _model
    from usdm_amd.voicebox.model import networks
    cfg, seed = SMALL_VB, 7
    m = _model(cfg, seed, dev)
    sd = VO.random_state_dict(cfg, seed)
    g = torch.Generator().manual_seed(3)
    nt, gs = 4, 1.0
    outs = {}
    for S, P in ((70, 20), (90, 33)):                       # S + 1 = 71, 91 -> both in the 96-row bucket
        assert m.estimator.bucket_frames(S) == 95
        x = torch.randint(0, cfg["n_tokens"], (1, S), generator=g)
        cond = torch.zeros(1, 80, S); cond[:, :, :P] = torch.randn(1, 80, P, generator=g)
        noise = [torch.randn(1, 80, S, generator=g) for _ in range(VO.noise_count(nt, "heun", True))]
        ref = VO.generate(sd, cfg, x, cond, torch.tensor([S]), nt, noise, "heun", gs, True, torch.tensor([P]))
        args = (x.to(dev), cond.to(dev), torch.tensor([S]).to(dev))
        kw = dict(n_timesteps=nt, solver="heun", gradient_scale=gs, speech_prompt=True, prompt_lengths=torch.tensor([P]).to(dev),
                  noise=torch.stack(noise))
        out = m.generate(*args, **kw)
        assert out.shape == ref.shape == (1, 80, S)
        r = _rel(out, ref)
        print(f"S={S}: bucketed generate rel L2 {r:.4f}")
        assert r <= 3e-2
        outs[S] = (args, kw, out)
    assert len(m.estimator._plans) == 1                     # ONE plan / graph / workspace served both lengths
    for S, (args, kw, out) in outs.items():                 # alternate again on the shared plan: bit-identical
        assert torch.equal(m.generate(*args, **kw), out)
    rnd = random.Random(1)
    for _ in range(50):
        S = rnd.randint(10, 700)
        o = m.generate(torch.zeros(1, S, dtype=torch.long, device=dev), torch.zeros(1, 80, S, device=dev), torch.tensor([S]).to(dev),
                       n_timesteps=1, solver="euler", gradient_scale=0.0, speech_prompt=False)
        assert o.shape == (1, 80, S) and torch.isfinite(o).all()
    assert len(m.estimator._plans) <= networks.MAX_PLANS and m.estimator._plans.evictions > 0


def test_bigvgan_two_lengths_share_a_workspace_and_cache_is_bounded(dev):
    from oracle import bigvgan_oracle as BO
    from tests.test_bigvgan_gpu import _model, _snr
    from usdm_amd.voicebox.vocoder import models as VM
    m, sd, h = _model(64, 11, dev)
    g = torch.Generator().manual_seed(5)
    mels = {T: torch.randn(1, 80, T, generator=g) * 2.1575 - 5.5419 for T in (37, 50)}      # both in the 64-frame bucket
    refs = {T: BO.bigvgan_forward(sd, dict(h), mel) for T, mel in mels.items()}
    first = {}
    for rep in range(3):                                    # 37, 50, 37, 50, ...: the workspace changes hands every call
        for T, mel in mels.items():
            wav = m(mel.to(dev))
            snr = _snr(wav.cpu(), refs[T])
            assert wav.shape == (1, 1, 256 * T) and snr >= 50.0, (T, rep, snr)
            if rep == 0:
                first[T] = wav.clone()
            else:
                assert torch.equal(wav, first[T])           # graph replay on a re-zeroed shared workspace: same bits
    assert len(m._plans) == 1
    arena, plans = next(iter(m._plans.values()))
    assert len(plans) == 2
    a, b = m(mels[37].to(dev)), m(mels[37].to(dev))
    assert a.data_ptr() != b.data_ptr()                     # forward returns a NEW tensor each call
    rnd = random.Random(2)
    for _ in range(50):
        T = rnd.randint(3, 400)
        w = m(torch.randn(1, 80, T, device=dev))
        assert w.shape == (1, 1, 256 * T) and torch.isfinite(w).all()
    assert len(m._plans) <= VM.MAX_ARENAS and all(len(p) <= VM.MAX_PLANS for _, p in m._plans.values())
    assert m._plans.evictions > 0


def test_unit_extractor_two_lengths_share_a_workspace_and_cache_is_bounded(dev):
    from oracle import w2v_oracle as WO
    from usdm_amd import unit_extractor as UE
    cfg = dict(WO.XLSR_1B, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_conv_pos_embedding_groups=4, n_units=300)
    sd = WO.random_state_dict(cfg, 1, n_layers=3)
    g = torch.Generator().manual_seed(4)
    cen = torch.randn(300, 256, generator=g)
    ue = UE.UnitExtractor(None, None, device=dev, config=cfg, state_dict=sd, centroids=cen)
    waves = {n: torch.randn(n, generator=g) * 0.1 for n in (9000, 15000)}                   # both in the 16000-sample bucket
    want = {}
    for n, w in waves.items():
        ids, dist = WO.kmeans_assign(WO.features(sd, cfg, w, 2), cen)
        want[n] = (ids, dist)
    for rep in range(2):
        for n, w in waves.items():
            got = ue.predict(w.to(dev), 2).cpu()
            ids, dist = want[n]
            top2 = torch.topk(dist, 2, largest=False).values
            bad = got != ids
            assert got.shape == ids.shape and float((~bad).float().mean()) >= 0.97
            assert bool(((top2[:, 1] - top2[:, 0])[bad] <= 1e-4 * dist.abs().max()).all())
    assert len(ue._plans) == 1
    rnd = random.Random(3)
    for _ in range(50):
        n = 2 * rnd.randint(400, 40000)
        assert ue.predict(torch.randn(n, device=dev) * 0.1, 2).numel() >= 1
    assert len(ue._plans) <= UE.MAX_ARENAS and all(len(p) <= UE.MAX_PLANS for _, p in ue._plans.values())
