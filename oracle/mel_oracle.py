"""Oracle: the reference-prompt mel front end restated on CPU torch/numpy (TEST INFRASTRUCTURE).

  mel_spectrogram   src/decoder/voicebox/vocoder/meldataset.py:55-78 (reflect pad, torch.stft, magnitude, mel, log)
  get_mel           src/decoder/voicebox/util/model_util.py:24-38 (resample, truncate, clamp)
Third-party pieces that are absent here and therefore **parity unpinned** (restated from their published algorithms):
  librosa.filters.mel (Slaney scale, norm='slaney')   meldataset.py:62
  torchaudio.transforms.Resample (sinc_interp_hann, lowpass_filter_width=6, rolloff=0.99)   model_util.py:27
The STFT part is torch.stft itself.
"""
import math

import numpy as np
import torch


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults (htk=False, norm='slaney') -> f32 [n_mels, 1+n_fft/2]."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, len(fftfreqs)))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return (w * enorm[:, None]).astype(np.float32)


def mel_spectrogram(y, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256, win_size=1024, fmin=0, fmax=8000):
    """y f32 [B, n] -> log-mel f32 [B, num_mels, frames] (center=False path of the reference)."""
    mel = torch.from_numpy(slaney_mel_filterbank(sampling_rate, n_fft, num_mels, fmin, fmax))
    pad = int((n_fft - hop_size) / 2)
    y = torch.nn.functional.pad(y.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(y, n_fft, hop_length=hop_size, win_length=win_size, window=torch.hann_window(win_size), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(torch.real(spec * spec.conj() + 1e-9))
    spec = torch.matmul(mel, spec)
    return torch.log(torch.clamp(spec, min=1e-5))


def resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.functional._get_sinc_resample_kernel (sinc_interp_hann), float64 -> f32 [new, 2*width+orig], width."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, :] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    k = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    return k.to(torch.float32), width, orig, new


def resample(x, orig_freq, new_freq):
    """x f32 [n] -> f32 [ceil(new*n/orig)] (torchaudio.functional.resample semantics)."""
    if orig_freq == new_freq:
        return x
    k, width, orig, new = resample_kernel(orig_freq, new_freq)
    n = x.numel()
    xp = torch.nn.functional.pad(x[None, None], (width, width + orig))
    y = torch.nn.functional.conv1d(xp, k[:, None, :], stride=orig)       # [1, new, frames]
    y = y.transpose(1, 2).reshape(-1)
    return y[: math.ceil(new * n / orig)]


def get_mel(audio, sr, length=None, sampling_rate=22050, **mel_kw):
    """model_util.py:24-38 on an already-decoded float waveform [n]."""
    if sr != sampling_rate:
        audio = resample(audio, sr, sampling_rate)
    if length:
        audio = audio[:length]
    audio = audio.clamp(-1, 1)
    return mel_spectrogram(audio[None], sampling_rate=sampling_rate, **mel_kw)
