"""Generates tests/golden/*.npz by running the REFERENCE's own classes (imported read-only from
/root/reference) on CPU in the build container.  Only inputs/outputs (data) are stored; weights are
re-derived from seeds by oracle.*.random_state_dict, so no reference source or bytecode is copied.

Harness-side shims (none touches reference code):
  * `librosa`, `torchaudio` are absent here and are only needed for *imports* of modules unrelated
    to the arithmetic captured below -> empty stand-in modules in sys.modules (never called);
  * networks.py:319 hard-codes `.cuda()` -> torch.Tensor.cuda is a no-op while the fixtures are made;
  * torch.randn_like is replaced by a queue so the noise draws are caller-supplied (voicebox.py:116,127,142).

Run:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src/decoder")

for name, attrs in {
    "librosa": {}, "librosa.util": {"normalize": None}, "librosa.filters": {"mel": None},
    "torchaudio": {}, "torchaudio.transforms": {"Resample": None},
}.items():
    if name not in sys.modules:
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
sys.modules["librosa"].load = None
torch.Tensor.cuda = lambda self, *a, **k: self

from oracle import bigvgan_oracle as BO  # noqa: E402
from oracle import voicebox_oracle as VO  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


class NoiseQueue:
    def __init__(self, noise):
        self.it = iter(noise)

    def __call__(self, like, **kw):
        n = next(self.it)
        assert n.shape == like.shape
        return n.clone()


def ref_voicebox(cfg, sd):
    from voicebox.model import Voicebox
    m = Voicebox(n_feats=cfg["n_feats"], n_tokens=cfg["n_tokens"], embedding_dim=cfg["embedding_dim"],
                 hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                 num_attention_heads=cfg["num_attention_heads"], num_hidden_layers=cfg["num_hidden_layers"],
                 convpos_width=cfg["convpos_width"], convpos_groups=cfg["convpos_groups"],
                 convpos_depth=cfg["convpos_depth"], attention_dropout=0.0, activation_dropout=0.1,
                 hidden_dropout=0.0, solver="euler", sigma_min=cfg["sigma_min"]).eval()
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return m


from tests.golden.configs import SMALL_VB  # noqa: E402


def make_voicebox():
    g = torch.Generator().manual_seed(100)
    # --- small width, full generate (heun + CFG + speech prompt; euler without prompt)
    cfg = SMALL_VB
    sd = VO.random_state_dict(cfg, seed=11)
    m = ref_voicebox(cfg, sd)
    S, P = 70, 22
    x = torch.randint(0, cfg["n_tokens"], (1, S), generator=g)
    cond = torch.randn(1, 80, S, generator=g)
    cond[:, :, P:] = 0
    lengths = torch.tensor([S])
    t = torch.tensor([0.37]).view(1, 1, 1)
    y = torch.randn(1, 80, S, generator=g)
    with torch.no_grad():
        est = m.estimator(x, y, cond, t, lengths)
        # batch of 2 with CFG-style doubling goes through the same call inside sample()
        nt = 6
        noise_h = [torch.randn(1, 80, S, generator=g) for _ in range(VO.noise_count(nt, "heun", True))]
        orig = torch.randn_like
        torch.randn_like = NoiseQueue(noise_h)
        gen_h = m.generate(x, cond, lengths, n_timesteps=nt, solver="heun", gradient_scale=1.0,
                           speech_prompt=True, prompt_lengths=torch.tensor([P]))
        noise_e = [torch.randn(1, 80, S, generator=g) for _ in range(VO.noise_count(3, "euler", False))]
        torch.randn_like = NoiseQueue(noise_e)
        gen_e = m.generate(x, torch.zeros_like(cond), lengths, n_timesteps=3, solver="euler", gradient_scale=0.7,
                           speech_prompt=False)
        torch.randn_like = orig
    save("voicebox_small.npz", seed=11, x=x, y=y, cond=cond, t=t, lengths=lengths, est=est, P=P,
         noise_h=torch.stack(noise_h), gen_h=gen_h, nt_h=nt, noise_e=torch.stack(noise_e), gen_e=gen_e, nt_e=3,
         gs_e=0.7)
    # --- full width (config.json), one CFG-doubled estimator call at short S
    cfg = VO.VOICEBOX_CFG
    sd = VO.random_state_dict(cfg, seed=12)
    m = ref_voicebox(cfg, sd)
    S = 45
    x = torch.randint(0, cfg["n_tokens"], (1, S), generator=g)
    xx = torch.cat([cfg["n_tokens"] * torch.ones_like(x), x], 0)
    y = torch.randn(1, 80, S, generator=g)
    cond = torch.randn(1, 80, S, generator=g)
    with torch.no_grad():
        est = m.estimator(xx, torch.cat([y, y], 0), torch.cat([torch.zeros_like(cond), cond], 0),
                          torch.full((2, 1, 1), 0.61), torch.tensor([S, S]))
    save("voicebox_full.npz", seed=12, x=xx, y=y, cond=cond, t=0.61, est=est)


def make_voicebox_ragged():
    """Batch of 2 with different lengths: the padding masks of networks.py:314-341 and the `* y_mask` products.
    (estimator only: CFM.sample views the scalar t as [B,1,1], so the reference's generate() is batch-1, voicebox.py:52)"""
    g = torch.Generator().manual_seed(101)
    cfg = SMALL_VB
    sd = VO.random_state_dict(cfg, seed=13)
    m = ref_voicebox(cfg, sd)
    S = 52
    lengths = torch.tensor([52, 33])
    x = torch.randint(0, cfg["n_tokens"], (2, S), generator=g)
    y = torch.randn(2, 80, S, generator=g)
    cond = torch.randn(2, 80, S, generator=g)
    t = torch.tensor([0.3, 0.8]).view(2, 1, 1)
    with torch.no_grad():
        est = m.estimator(x, y, cond, t, lengths)
    save("voicebox_ragged.npz", seed=13, x=x, y=y, cond=cond, t=t, lengths=lengths, est=est)


def ref_bigvgan(h, sd):
    from voicebox.vocoder.env import AttrDict
    from voicebox.vocoder.models import BigVGAN
    m = BigVGAN(AttrDict(h)).eval()
    m.remove_weight_norm()
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert all(k.endswith("filter") for k in res.missing_keys), res.missing_keys
    return m


def make_bigvgan():
    g = torch.Generator().manual_seed(200)
    # filter taps the reference registers as buffers (pins kaiser_sinc_filter1d restatement)
    from voicebox.vocoder.alias_free_torch.filter import kaiser_sinc_filter1d
    taps = kaiser_sinc_filter1d(0.25, 0.3, 12).view(-1)
    # Activation1d alone
    from voicebox.vocoder.alias_free_torch.act import Activation1d
    from voicebox.vocoder.activations import SnakeBeta
    C, T = 6, 37
    act = Activation1d(SnakeBeta(C, alpha_logscale=True))
    with torch.no_grad():
        act.act.alpha.copy_(torch.randn(C, generator=g) * 0.5)
        act.act.beta.copy_(torch.randn(C, generator=g) * 0.5)
        xa = torch.randn(1, C, T, generator=g) * 2
        ya = act(xa)
    save("bigvgan_act.npz", taps=taps, x=xa, alpha=act.act.alpha, beta=act.act.beta, y=ya)
    # narrow generator (initial channel 64) on 12 frames; full-width generator on 6 frames
    for name, c0, T, seed in (("bigvgan_small.npz", 64, 12, 21), ("bigvgan_full.npz", 1536, 6, 22)):
        h = dict(BO.BIGVGAN_22K_80, upsample_initial_channel=c0)
        sd = BO.random_state_dict(h, seed=seed)
        m = ref_bigvgan(h, sd)
        mel = torch.randn(1, 80, T, generator=g) * 2.1575 - 5.5419
        with torch.no_grad():
            wav = m(mel)
        save(name, seed=seed, c0=c0, mel=mel, wav=wav)


def make_bigvgan_variant():
    """resblock '2' (AMPBlock2) + 'snake' activation: the other generator configuration the reference supports."""
    g = torch.Generator().manual_seed(201)
    h = dict(BO.BIGVGAN_22K_80, upsample_initial_channel=64, resblock="2", activation="snake",
             resblock_dilation_sizes=[[1, 3], [1, 3], [1, 3]])
    sd = BO.random_state_dict(h, seed=23)
    m = ref_bigvgan(h, sd)
    mel = torch.randn(1, 80, 10, generator=g) * 2.1575 - 5.5419
    with torch.no_grad():
        wav = m(mel)
    save("bigvgan_amp2_snake.npz", seed=23, c0=64, mel=mel, wav=wav)


def make_process_unit():
    from voicebox.util.model_util import process_unit
    from voicebox.vocoder.env import AttrDict
    g = torch.Generator().manual_seed(300)
    hps = AttrDict(sampling_rate=22050, hop_size=256)
    outs = {}
    for i, n in enumerate((1, 7, 149, 500)):
        u = torch.randint(0, 10000, (n,), generator=g)
        o, new_len = process_unit(u, hps, "cpu")
        outs[f"u{i}"] = u
        outs[f"o{i}"] = o
        outs[f"len{i}"] = new_len
    # adversarial: repeated ids and descending ids exercise the mode tie rule
    u = torch.tensor([5, 5, 3, 9, 9, 9, 2, 1, 0, 7, 7, 4] * 5)
    o, new_len = process_unit(u, hps, "cpu")
    outs["u4"], outs["o4"], outs["len4"] = u, o, new_len
    save("process_unit.npz", **outs)


if __name__ == "__main__":
    if "--only-variant" in sys.argv:
        make_bigvgan_variant()
        sys.exit(0)
    if "--only-ragged" in sys.argv:
        make_voicebox_ragged()
        sys.exit(0)
    make_process_unit()
    make_bigvgan_variant()
    make_bigvgan()
    make_voicebox()
    make_voicebox_ragged()
