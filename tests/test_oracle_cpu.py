"""CPU: the oracle restatements reproduce the golden vectors produced by the reference itself."""
import os

import numpy as np
import torch

from oracle import bigvgan_oracle as BO
from oracle import units_oracle as UO
from oracle import voicebox_oracle as VO

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(n):
    return {k: torch.from_numpy(v) if v.ndim else v.item() for k, v in np.load(os.path.join(G, n)).items()}


def _close(a, b, tol):
    err = (a - b).abs().max().item()
    assert err <= tol * (b.abs().max().item() + 1e-12), err


def test_filter_taps_and_activation1d():
    d = _load("bigvgan_act.npz")
    taps = BO.aa_filter_taps()
    assert torch.allclose(taps, d["taps"], atol=1e-7)
    y = BO.activation1d(d["x"], d["alpha"], d["beta"], taps)
    _close(y, d["y"], 1e-6)


def test_bigvgan_small_and_full():
    for name in ("bigvgan_small.npz", "bigvgan_full.npz"):
        d = _load(name)
        h = dict(BO.BIGVGAN_22K_80, upsample_initial_channel=int(d["c0"]))
        sd = BO.random_state_dict(h, seed=int(d["seed"]))
        wav = BO.bigvgan_forward(sd, h, d["mel"])
        assert wav.shape == d["wav"].shape
        _close(wav, d["wav"], 1e-5)
        assert d["wav"].abs().mean() > 0.05  # fixture is not degenerate
    d = _load("bigvgan_amp2_snake.npz")   # AMPBlock2 + Snake variant
    h = dict(BO.BIGVGAN_22K_80, upsample_initial_channel=64, resblock="2", activation="snake", resblock_dilation_sizes=[[1, 3]] * 3)
    _close(BO.bigvgan_forward(BO.random_state_dict(h, seed=int(d["seed"])), h, d["mel"]), d["wav"], 1e-5)


def test_voicebox_small():
    from tests.golden.configs import SMALL_VB
    d = _load("voicebox_small.npz")
    cfg = SMALL_VB
    sd = VO.random_state_dict(cfg, seed=int(d["seed"]))
    est = VO.estimator_forward(sd, cfg, d["x"], d["y"], d["cond"], d["t"], d["lengths"])
    _close(est, d["est"], 2e-5)
    P = int(d["P"])
    gen = VO.generate(sd, cfg, d["x"], d["cond"], d["lengths"], int(d["nt_h"]), list(d["noise_h"]), "heun", 1.0, True,
                      torch.tensor([P]))
    _close(gen, d["gen_h"], 1e-4)
    gen = VO.generate(sd, cfg, d["x"], torch.zeros_like(d["cond"]), d["lengths"], int(d["nt_e"]), list(d["noise_e"]),
                      "euler", float(d["gs_e"]), False)
    _close(gen, d["gen_e"], 1e-4)


def test_voicebox_ragged_batch():
    from tests.golden.configs import SMALL_VB
    d = _load("voicebox_ragged.npz")
    sd = VO.random_state_dict(SMALL_VB, seed=int(d["seed"]))
    est = VO.estimator_forward(sd, SMALL_VB, d["x"], d["y"], d["cond"], d["t"], d["lengths"])
    _close(est, d["est"], 2e-5)
    assert est[1, :, int(d["lengths"][1]):].abs().max() == 0


def test_voicebox_full_width():
    d = _load("voicebox_full.npz")
    cfg = VO.VOICEBOX_CFG
    sd = VO.random_state_dict(cfg, seed=int(d["seed"]))
    S = d["y"].shape[-1]
    est = VO.estimator_forward(sd, cfg, d["x"], torch.cat([d["y"]] * 2), torch.cat([torch.zeros_like(d["cond"]), d["cond"]]),
                               torch.full((2, 1, 1), float(d["t"])), torch.tensor([S, S]))
    _close(est, d["est"], 5e-5)


def test_process_unit():
    d = np.load(os.path.join(G, "process_unit.npz"))
    for i in range(5):
        out, new_len = UO.process_unit(d[f"u{i}"].tolist())
        assert new_len == int(d[f"len{i}"])
        assert out == d[f"o{i}"][0].tolist()
    # frame counts the survey quotes: 500 -> 861, 149 -> 256
    assert d["o3"].shape[1] == 861 and d["o2"].shape[1] == 256


def test_mistral_oracle_vs_installed_transformers():
    """The LLM arithmetic of the reference is third-party HF code; pin the restatement against the
    transformers build present in this image (version skew vs the reference's 4.40.2 is documented)."""
    import pytest
    transformers = pytest.importorskip("transformers")
    from oracle import mistral_oracle as MO
    cfg = dict(MO.MISTRAL_7B_USDM, vocab_size=300, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
               num_attention_heads=4, num_key_value_heads=2, head_dim=64)
    hf_cfg = transformers.MistralConfig(
        vocab_size=cfg["vocab_size"], hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
        num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
        num_key_value_heads=cfg["num_key_value_heads"], head_dim=cfg["head_dim"], rms_norm_eps=cfg["rms_norm_eps"],
        rope_theta=cfg["rope_theta"], max_position_embeddings=4096, sliding_window=4096, attn_implementation="eager",
        tie_word_embeddings=False)
    ids = torch.randint(0, 300, (1, 17), generator=torch.Generator().manual_seed(1))
    for dtype, tol in ((torch.float32, 2e-5), (torch.bfloat16, 3e-2)):
        sd = MO.random_state_dict(cfg, seed=3, dtype=dtype)
        m = transformers.MistralForCausalLM(hf_cfg).to(dtype).eval()
        m.load_state_dict(sd, strict=True)
        with torch.no_grad():
            ref = m(ids).logits[0].float()
        got, _ = MO.forward(sd, cfg, ids[0])
        _close(got, ref, tol)
    # greedy generate with a ban mask == HF generate(do_sample=True, top_k=1) on the same model (fp32: no ties)
    sd = MO.random_state_dict(cfg, seed=3, dtype=torch.float32)
    m = transformers.MistralForCausalLM(hf_cfg).eval()
    m.load_state_dict(sd, strict=True)
    bad = [[i] for i in range(0, 150)]
    with torch.no_grad():
        hf = m.generate(input_ids=ids, max_length=17 + 12, do_sample=True, top_k=1, top_p=1.0, temperature=1.0,
                        bad_words_ids=bad, eos_token_id=299, pad_token_id=0)
    mine = MO.greedy_generate(sd, cfg, ids[0], 12, bad_words_ids=bad, eos_token_id=299)
    assert hf[0].tolist() == mine
    # sliding window shorter than the context (Mistral-7B-v0.1: 4096; reference src/model.py:337-371): prefill logits and generate
    hf_cfg.sliding_window = 8
    wcfg = dict(cfg, sliding_window=8)
    m = transformers.MistralForCausalLM(hf_cfg).eval()
    m.load_state_dict(sd, strict=True)
    ids = torch.randint(0, 300, (1, 21), generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        ref = m(ids).logits[0].float()
        hf = m.generate(input_ids=ids, max_length=21 + 14, do_sample=True, top_k=1, top_p=1.0, temperature=1.0,
                        bad_words_ids=bad, eos_token_id=299, pad_token_id=0)
    _close(MO.forward(sd, wcfg, ids[0])[0], ref, 2e-5)
    assert (MO.forward(sd, cfg, ids[0])[0] - ref).abs().max() > 0.1          # the window matters on this input
    assert hf[0].tolist() == MO.greedy_generate(sd, wcfg, ids[0], 14, bad_words_ids=bad, eos_token_id=299)


def test_w2v_oracle_vs_hf_wav2vec2_and_kmeans():
    """Structural pin of the XLS-R restatement against HF Wav2Vec2Model (independent implementation of the
    same architecture); the seamless_communication tokenizer itself is absent -> 'parity unpinned'."""
    import pytest
    transformers = pytest.importorskip("transformers")
    from oracle import w2v_oracle as WO
    cfg = dict(WO.XLSR_1B, conv_dim=(32,) * 7, hidden_size=64, num_attention_heads=4, intermediate_size=128,
               num_hidden_layers=3, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, n_units=50)
    hf = transformers.Wav2Vec2Config(
        hidden_size=64, num_hidden_layers=3, num_attention_heads=4, intermediate_size=128, conv_dim=cfg["conv_dim"],
        conv_kernel=cfg["conv_kernel"], conv_stride=cfg["conv_stride"], feat_extract_norm="layer", conv_bias=True,
        do_stable_layer_norm=True, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4, hidden_dropout=0.0,
        attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0,
        attn_implementation="eager")
    m = transformers.Wav2Vec2Model(hf).eval()
    sd = WO.random_state_dict(cfg, seed=4)
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k in ("masked_spec_embed", "encoder.layer_norm.weight", "encoder.layer_norm.bias") for k in res.missing_keys), res.missing_keys
    wave = torch.randn(4000, generator=torch.Generator().manual_seed(5)) * 0.1
    normed = torch.nn.functional.layer_norm(wave, wave.shape)
    with torch.no_grad():
        hs = m(normed[None], output_hidden_states=True).hidden_states
    for idx in (0, 1, 2):  # hidden_states[idx+1] is the output of encoder layer idx
        got = WO.features(sd, cfg, wave, idx)
        assert got.shape[0] == WO.n_frames(4000, cfg)
        if idx < 2:
            _close(got, hs[idx + 1][0], 2e-5)
    assert WO.n_frames(160000, WO.XLSR_1B) == 499 and WO.n_frames(48000, WO.XLSR_1B) == 149
    # k-means stage: formula == brute-force fp64 nearest centroid
    x = torch.randn(40, 64, generator=torch.Generator().manual_seed(6))
    C = torch.randn(50, 64, generator=torch.Generator().manual_seed(7))
    ids, _ = WO.kmeans_assign(x, C)
    brute = ((x.double()[:, None] - C.double()[None]) ** 2).sum(-1).argmin(-1)
    assert torch.equal(ids, brute)


def test_mel_oracle_sanity():
    """librosa/torchaudio are absent (parity unpinned): check the restated pieces against their defining properties."""
    from oracle import mel_oracle as MO
    fb = MO.slaney_mel_filterbank(22050, 1024, 80, 0, 8000)
    assert fb.shape == (80, 513) and (fb >= 0).all()
    freqs = np.linspace(0, 11025, 513)
    assert fb[:, freqs > 8000.0 + 1e-6].sum() == 0           # nothing above fmax
    areas = (fb * (freqs[1] - freqs[0])).sum(1)
    assert np.allclose(areas, 1.0, atol=0.06)                 # Slaney normalisation: ~unit area per filter
    # resampler: a sine well below both Nyquists keeps its amplitude and frequency
    t = torch.arange(16000) / 16000.0
    y = MO.resample(torch.sin(2 * torch.pi * 440 * t), 16000, 22050)
    t2 = torch.arange(y.numel()) / 22050.0
    assert y.numel() == 22050 and (y[200:-200] - torch.sin(2 * torch.pi * 440 * t2)[200:-200]).abs().max() < 2e-2


def test_sampling_oracle_matches_hf_warpers():
    """oracle/sampling_oracle.filtered_probs keeps exactly the ids HF's warpers keep (tie-free inputs) with the same
    probabilities; the Philox stream is the published Philox4x32-10 (known-answer vector of Random123)."""
    import torch
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    from oracle import sampling_oracle as so
    g = np.random.default_rng(0)
    for V, T, k, p in ((1000, 0.7, 50, 0.9), (42003, 1.3, 0, 0.95), (42003, 1.0, 200, 1.0), (517, 0.5, 10, 0.5)):
        x = (g.standard_normal(V) * 3).astype(np.float32)
        x[g.integers(0, V, V // 10)] = -np.inf                    # banned ids
        s = torch.from_numpy(x)[None]
        ids = torch.zeros(1, 1, dtype=torch.long)
        s = TemperatureLogitsWarper(T)(ids, s)
        if k:
            s = TopKLogitsWarper(k)(ids, s)
        if p < 1:
            s = TopPLogitsWarper(p)(ids, s)
        ref = torch.softmax(s[0].double(), -1).numpy()
        mine = so.filtered_probs(x, T, k, p)
        assert ((ref > 0) == (mine > 0)).all(), (V, T, k, p)
        np.testing.assert_allclose(mine, ref, rtol=2e-6, atol=1e-9)
    # Random123 known-answer test: philox4x32-10, counter = key = 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8
    c, kk = [0, 0, 0, 0], [0, 0]
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c[0], 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ kk[0]) & so.M32, p1 & so.M32, ((p0 >> 32) ^ c[3] ^ kk[1]) & so.M32, p0 & so.M32]
        kk = [(kk[0] + 0x9E3779B9) & so.M32, (kk[1] + 0xBB67AE85) & so.M32]
    assert c == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert so.philox_uniform(0, 0) == float(((0x6627E8D5 >> 5) << 26) | (0xE169C58D >> 6)) / 2 ** 53
    # the draw lands in the interval of the returned id
    tok, pr = so.sample(x, step=7, temperature=0.8, top_k=40, top_p=0.9, seed=123)
    cs = np.cumsum(pr)
    u = so.philox_uniform(123, 7)
    assert (cs[tok - 1] if tok else 0.0) <= u * cs[-1] < cs[tok] and pr[tok] > 0
