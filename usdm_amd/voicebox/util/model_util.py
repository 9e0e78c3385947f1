"""Decoder orchestration drop-in (reference: src/decoder/voicebox/util/model_util.py:18-105).

Same names and signatures: mel_mean, mel_std, process_unit, initialize_decoder, reconstruct_speech.
Differences forced by the environment (no network, no librosa/torchaudio):
  * initialize_decoder loads from local directories under model_cache_dir
    (<dir>/xlsr-token-Voicebox, <dir>/bigvgan_22khz_80band) instead of hub names;
  * the reference-prompt mel front end (get_mel, model_util.py:24-38) is SURVEY.md §8(f) "next":
    reconstruct_speech accepts `reference_mel=` / `reference_unit=` tensors instead of decoding a file.
"""
import os

import torch

from ... import ops
from ..model import Voicebox
from ..vocoder.meldataset import mel_spectrogram, resample
from ..vocoder.models import BigVGAN

mel_mean = -5.5419
mel_std = 2.1575


def load_wav_to_torch(full_path):
    """util/train_util.py:36-38 (torchaudio.load) with scipy: mono float waveform in [-1, 1] + sample rate (host I/O)."""
    import numpy as np
    from scipy.io.wavfile import read
    sr, data = read(full_path)
    x = data.astype(np.float32)
    if data.dtype == np.int16:
        x /= 32768.0
    elif data.dtype == np.int32:
        x /= 2147483648.0
    if x.ndim == 2:
        x = x[:, 0]
    return torch.from_numpy(x), sr


def get_mel(filepath, length=None, hps=None, device="cuda"):
    """wav file -> log-mel [1, num_mels, frames] on the GPU (model_util.py:24-38)."""
    audio, sr = load_wav_to_torch(filepath)
    audio = audio.to(device)
    if sr != hps.sampling_rate:
        audio = resample(audio, sr, hps.sampling_rate)
    if length:
        audio = audio[:length]
    # clamp(-1, 1) of float audio happens inside the framing kernel
    return mel_spectrogram(audio.unsqueeze(0), hps.n_fft, hps.num_mels, hps.sampling_rate, hps.hop_size, hps.win_size, hps.fmin,
                           hps.fmax, center=False)


def process_unit(unit, hps, device):
    """50 Hz unit ids -> one id per mel frame (mode over each hop), on the GPU (model_util.py:50-54)."""
    unit = unit.to(device)
    rep = hps.sampling_rate // 50
    out = ops.process_unit(unit, rep, hps.hop_size)
    new_length = (unit.numel() * rep) // hps.hop_size * hps.hop_size
    return out.unsqueeze(0), new_length


def initialize_decoder(model_cache_dir, device):
    """model_util.py:57-69: both models .eval(), the vocoder's weight norm removed.  The hub names resolve inside
    model_cache_dir (huggingface_hub cache layout or <dir>/<name>/; usdm_amd/checkpoints.py)."""
    from ...checkpoints import resolve_local
    voicebox = Voicebox.from_pretrained(resolve_local(model_cache_dir, "naver-ai/xlsr-token-Voicebox", ("config.json",))).to(device).eval()
    vocoder = BigVGAN.from_pretrained(resolve_local(model_cache_dir, "nvidia/bigvgan_22khz_80band", ("config.json",))).to(device).eval()
    vocoder.remove_weight_norm()
    return voicebox, vocoder


@torch.inference_mode()
def reconstruct_speech(agent_unit, device, reference_path, token_extractor, voicebox, vocoder, n_timesteps=50,
                       reference_mel=None, reference_unit=None, noise=None, cfg_group=None):
    """units -> waveform float32 numpy [256 * frames] (model_util.py:72-105).

    With a speech prompt, supply `reference_unit` (50 Hz ids of the prompt, e.g. from
    token_extractor.predict) and `reference_mel` ([1, 80, frames] log-mel, un-normalised).
    cfg_group: see Voicebox.generate (the two CFG halves on two ranks; both ranks must then pass the same `noise`)."""
    agent_unit, _ = process_unit(agent_unit, vocoder.h, device)
    if reference_path is not None and reference_mel is None:
        # reference prompt from a wav file, as the reference does (model_util.py:76-82); file decode + 16 kHz resampling are host I/O
        from ...inference import load_audio_16k
        ref16 = torch.as_tensor(load_audio_16k(reference_path), dtype=torch.float32).to(device)
        reference_unit = token_extractor.predict(ref16, 35 - 1)
        _, new_length = process_unit(reference_unit, vocoder.h, device)
        reference_mel = get_mel(reference_path, new_length, vocoder.h, device=device)
    if reference_mel is not None:
        if reference_unit is None:
            raise ValueError("reference_unit (50 Hz ids of the prompt) is required with reference_mel")
        reference_unit, new_length = process_unit(reference_unit, vocoder.h, device)
        P = reference_unit.shape[-1]
        reference_mel = (reference_mel.to(device).float()[:, :, :P] - mel_mean) / mel_std
        dummy_y = torch.zeros(agent_unit.shape[0], vocoder.h.num_mels, P + agent_unit.shape[-1], device=device)
        dummy_y[:, :, :P] = reference_mel
        dummy_y_lengths = torch.LongTensor([dummy_y.shape[-1]]).to(device)
        prompt_lengths = torch.LongTensor([P]).to(device)
        unit = torch.cat([reference_unit, agent_unit], dim=-1)
        y_dec = voicebox.generate(unit, dummy_y, dummy_y_lengths, n_timesteps=n_timesteps, solver="heun", gradient_scale=1.0,
                                  speech_prompt=True, prompt_lengths=prompt_lengths, noise=noise, cfg_group=cfg_group)
        y_dec = y_dec[:, :, P:]
    else:
        dummy_y = torch.zeros(agent_unit.shape[0], vocoder.h.num_mels, agent_unit.shape[-1], device=device)
        dummy_y_lengths = torch.LongTensor([dummy_y.shape[-1]]).to(device)
        y_dec = voicebox.generate(agent_unit, dummy_y, dummy_y_lengths, n_timesteps=n_timesteps, solver="heun",
                                  gradient_scale=1.0, speech_prompt=False, noise=noise, cfg_group=cfg_group)
    # inverse normalisation is folded into the vocoder's layout kernel (y*std + mean, model_util.py:103)
    audio_dec = vocoder.forward(y_dec.contiguous(), mel_std, mel_mean).cpu().squeeze().clamp(-1, 1).numpy()
    return audio_dec
