"""Mel front end drop-in (reference: src/decoder/voicebox/vocoder/meldataset.py:55-78) on the GPU.

mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
  usdm_stft_frames (reflect pad + framing + Hann) -> usdm_gemm F32 against a host-built DFT matrix (exact-f32 MFMA)
  -> usdm_stft_mag sqrt(re^2+im^2+1e-9) -> usdm_gemm F32 with the Slaney mel filterbank, log(clamp(.,1e-5)) epilogue,
  transposed store -> [1, num_mels, frames].
`resample(x, orig, new)` is torchaudio.transforms.Resample (sinc_interp_hann, width 6, rolloff 0.99) as a polyphase
GEMM.  librosa/torchaudio are absent offline: the filterbank and resampling kernel are computed here on the host from
their published formulas (parity unpinned).
"""
import math

import numpy as np
import torch

from ... import ops
from ..._lib import ACT_LOGCLAMP

_cache = {}


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0), lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3) * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """Slaney-scale, area-normalised triangular filters (what librosa.filters.mel returns by default)."""
    freqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    pts = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    d = np.diff(pts)
    ramps = pts[:, None] - freqs[None, :]
    lower = -ramps[:-2] / d[:-1, None]
    upper = ramps[2:] / d[1:, None]
    w = np.maximum(0, np.minimum(lower, upper)) * (2.0 / (pts[2:] - pts[:-2]))[:, None]
    return w.astype(np.float32)


def _consts(n_fft, num_mels, sr, win_size, fmin, fmax, dev):
    key = (n_fft, num_mels, sr, win_size, fmin, fmax, dev.index)
    if key not in _cache:
        nb = n_fft // 2 + 1
        k = torch.arange(nb, dtype=torch.float64)[:, None] * torch.arange(n_fft, dtype=torch.float64)[None, :]
        ang = 2 * math.pi * k / n_fft
        dft = torch.cat([torch.cos(ang), -torch.sin(ang)], 0).to(torch.float32)           # [2*nb, n_fft]
        nbp = (nb + 15) // 16 * 16
        mel = torch.zeros(num_mels, nbp)
        mel[:, :nb] = torch.from_numpy(mel_filterbank(sr, n_fft, num_mels, fmin, fmax))
        _cache[key] = dict(dft=dft.to(dev).contiguous(), mel=mel.to(dev).contiguous(), win=torch.hann_window(win_size).to(dev), nb=nb, nbp=nbp)
    return _cache[key]


@torch.no_grad()
def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    if not y.is_cuda:
        raise RuntimeError("mel_spectrogram (usdm_amd) runs on the MI355X only; there is no CPU fallback")
    if center or win_size != n_fft or n_fft % 16:
        raise NotImplementedError("only center=False, win_size == n_fft (the reference's call, model_util.py:36-37)")
    if y.dim() != 2 or y.shape[0] != 1:
        raise ValueError("expected audio of shape [1, n]")
    dev = y.device
    c = _consts(n_fft, num_mels, sampling_rate, win_size, fmin, fmax, dev)
    x = y[0].contiguous().float()
    n = x.numel()
    pad = int((n_fft - hop_size) / 2)
    T = 1 + (n + 2 * pad - n_fft) // hop_size
    if T < 1 or n <= pad:
        raise ValueError("audio too short for one STFT frame")
    frames = torch.empty(T, n_fft, device=dev)
    ops.stft_frames(x, c["win"], frames, n=n, n_fft=n_fft, hop=hop_size, pad=pad, T=T)
    nb, nbp = c["nb"], c["nbp"]
    ldri = (2 * nb + 3) // 4 * 4
    ri = torch.empty(T, ldri, device=dev)
    ops.gemm(frames, c["dft"], M=T, N=2 * nb, Kc=n_fft, out32=ri, ldc=ldri)
    mag = torch.empty(T, nbp, device=dev)
    ops.stft_mag(ri, mag, ld=ldri, T=T, nbins=nb, eps=1e-9, ldo=nbp, nbins_pad=nbp)
    out = torch.empty(1, num_mels, T, device=dev)
    ops.gemm(mag, c["mel"], M=T, N=num_mels, Kc=nbp, act=ACT_LOGCLAMP, out32=out, ldc=T, transpose_out=True)
    return out


def _resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
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


@torch.no_grad()
def resample(x, orig_freq, new_freq):
    """x f32 [n] on the GPU -> [ceil(new*n/orig)] (torchaudio.transforms.Resample defaults)."""
    if orig_freq == new_freq:
        return x
    if not x.is_cuda:
        raise RuntimeError("resample (usdm_amd) runs on the MI355X only; there is no CPU fallback")
    dev = x.device
    key = ("rs", int(orig_freq), int(new_freq), dev.index)
    if key not in _cache:
        k, width, orig, new = _resample_kernel(orig_freq, new_freq)
        kw = k.shape[1]
        kc = (kw + 15) // 16 * 16
        W = torch.zeros(new, kc)
        W[:, :kw] = k
        _cache[key] = (W.to(dev).contiguous(), width, orig, new, kc)
    W, width, orig, new, kc = _cache[key]
    x = x.contiguous().float()
    n = x.numel()
    T = (n + 2 * width + orig - (2 * width + orig)) // orig + 1      # conv1d output length over the padded signal
    frames = torch.empty(T, kc, device=dev)
    ops.frame_signal(x, frames, n=n, frame_len=kc, hop=orig, offset=width, T=T)
    out = torch.empty(T, new, device=dev)
    ops.gemm(frames, W, M=T, N=new, Kc=kc, out32=out, ldc=new)
    return out.reshape(-1)[: math.ceil(new * n / orig)]
