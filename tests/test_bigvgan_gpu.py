"""GPU parity of the BigVGAN drop-in (HIP path through the C-ABI) against the reference's golden
vectors and the CPU oracle.
Tolerance (default path = exact-f32 MFMA, fp32 everywhere like the reference): max-abs error
<= 2e-3 on the tanh-bounded waveform and SNR >= 50 dB; the residual error is fp32 summation order
amplified by 109 chained sin^2 activations on random weights.
The bf16-operand variant is a measured trade: SNR >= 15 dB on these random-weight fixtures."""
import os
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
warnings.filterwarnings("ignore", category=FutureWarning)


def _snr(out, ref):
    return 10 * torch.log10(ref.pow(2).sum() / (out - ref).pow(2).sum()).item()


def _model(c0, seed, dev, dtype=torch.float32):
    from oracle import bigvgan_oracle as BO
    from usdm_amd.voicebox.vocoder.env import AttrDict
    from usdm_amd.voicebox.vocoder.models import BigVGAN
    h = AttrDict(dict(BO.BIGVGAN_22K_80, upsample_initial_channel=c0))
    m = BigVGAN(h, compute_dtype=dtype)
    m.remove_weight_norm()
    sd = BO.random_state_dict(h, seed)
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and all(k.endswith("filter") for k in res.missing_keys)
    return m.to(dev).eval(), sd, h


@pytest.mark.parametrize("name", ["bigvgan_small.npz", "bigvgan_full.npz"])
def test_bigvgan_golden(dev, name):
    d = np.load(os.path.join(G, name))
    m, _, _ = _model(int(d["c0"]), int(d["seed"]), dev)
    wav = m(torch.from_numpy(d["mel"]).to(dev)).cpu()
    ref = torch.from_numpy(d["wav"])
    assert wav.shape == ref.shape
    snr = _snr(wav, ref)
    err = (wav - ref).abs().max().item()
    print(name, "snr", snr, "maxerr", err)
    assert snr >= 50.0 and err <= 2e-3, (snr, err)
    mb, _, _ = _model(int(d["c0"]), int(d["seed"]), dev, torch.bfloat16)
    snr_b = _snr(mb(torch.from_numpy(d["mel"]).to(dev)).cpu(), ref)
    print(name, "bf16 snr", snr_b)
    assert snr_b >= 15.0


def test_bigvgan_vs_oracle_longer_and_batch(dev):
    from oracle import bigvgan_oracle as BO
    m, sd, h = _model(128, 5, dev)
    mel = torch.randn(2, 80, 53, generator=torch.Generator().manual_seed(9)) * 2.1575 - 5.5419
    ref = BO.bigvgan_forward(sd, h, mel)
    wav = m(mel.to(dev)).cpu()
    assert wav.shape == ref.shape == (2, 1, 53 * 256)
    assert _snr(wav, ref) >= 50.0
    # de-normalisation folded into the layout kernel (model_util.py:103)
    wav2 = m((mel[:1].to(dev) + 5.5419) / 2.1575, 2.1575, -5.5419).cpu()
    assert _snr(wav2, ref[:1]) >= 50.0


def test_bigvgan_rejects_cpu():
    from oracle import bigvgan_oracle as BO
    from usdm_amd.voicebox.vocoder.env import AttrDict
    from usdm_amd.voicebox.vocoder.models import BigVGAN
    m = BigVGAN(AttrDict(dict(BO.BIGVGAN_22K_80, upsample_initial_channel=64)))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 80, 4))


def test_bigvgan_amp2_snake_variant_golden(dev):
    """The other generator configuration the reference supports (models.py:153, 88-128; activations.py:9-59)."""
    from oracle import bigvgan_oracle as BO
    from usdm_amd.voicebox.vocoder.env import AttrDict
    from usdm_amd.voicebox.vocoder.models import BigVGAN
    d = np.load(os.path.join(G, "bigvgan_amp2_snake.npz"))
    h = AttrDict(dict(BO.BIGVGAN_22K_80, upsample_initial_channel=64, resblock="2", activation="snake",
                      resblock_dilation_sizes=[[1, 3], [1, 3], [1, 3]]))
    m = BigVGAN(h)
    m.remove_weight_norm()
    res = m.load_state_dict(BO.random_state_dict(h, int(d["seed"])), strict=False)
    assert not res.unexpected_keys and all(k.endswith("filter") for k in res.missing_keys)
    wav = m.to(dev).eval()(torch.from_numpy(d["mel"]).to(dev)).cpu()
    ref = torch.from_numpy(d["wav"])
    assert _snr(wav, ref) >= 50.0 and (wav - ref).abs().max().item() <= 2e-3
