"""GPU parity of the speech-prompt mel front end (mel_spectrogram, resample, get_mel) against the CPU oracle
(torch.stft + restated Slaney filterbank / sinc-Hann resampler).  fp32 kernels: log-mel within 2e-3 absolute
(the log of near-zero bins amplifies fp32 DFT rounding), resampled audio within 1e-5."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _audio(n, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 22050.0
    return (0.3 * torch.sin(2 * torch.pi * 220 * t) + 0.2 * torch.sin(2 * torch.pi * 3300 * t + 1.0) + 0.05 * torch.randn(n, generator=g)).float()


@pytest.mark.parametrize("n", [65536, 22050, 1500])
def test_mel_spectrogram(dev, n):
    from oracle import mel_oracle as MO
    from usdm_amd.voicebox.vocoder.meldataset import mel_spectrogram
    y = _audio(n, 1) * 1.7   # exceeds [-1, 1]: the reference clamps before the STFT (model_util.py:32)
    ref = MO.mel_spectrogram(y.clamp(-1, 1)[None])
    out = mel_spectrogram(y[None].to(dev), 1024, 80, 22050, 256, 1024, 0, 8000, center=False).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 2e-3


@pytest.mark.parametrize("orig,new", [(16000, 22050), (22050, 16000), (44100, 22050)])
def test_resample(dev, orig, new):
    from oracle import mel_oracle as MO
    from usdm_amd.voicebox.vocoder.meldataset import resample
    x = _audio(12345, 2)
    ref = MO.resample(x, orig, new)
    out = resample(x.to(dev), orig, new).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= 1e-5


def test_get_mel_from_wav_file(dev, tmp_path):
    from scipy.io.wavfile import write
    from oracle import mel_oracle as MO
    from usdm_amd.voicebox.util.model_util import get_mel
    from usdm_amd.voicebox.vocoder.env import AttrDict
    hps = AttrDict(sampling_rate=22050, n_fft=1024, num_mels=80, hop_size=256, win_size=1024, fmin=0, fmax=8000)
    x = _audio(48000, 3).clamp(-1, 1)
    p = os.path.join(tmp_path, "ref.wav")
    write(p, 16000, x.numpy())          # float32 wav at 16 kHz -> resampled to 22.05 kHz inside get_mel
    out = get_mel(p, length=65536, hps=hps, device=dev).cpu()
    ref = MO.get_mel(x, 16000, length=65536)
    assert out.shape == ref.shape == (1, 80, 256)
    assert (out - ref).abs().max().item() <= 2e-3
