"""GPU integration: the drop-in API surface end to end with reduced-size models — inference.sample() (three LLM rounds,
regex, reconstruct_speech with a reference wav: tokenizer -> process_unit -> get_mel -> Voicebox Heun/CFG/prompt -> BigVGAN)."""
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class ToyTokenizer:
    """Stand-in for the HF tokenizer of naver-ai/USDM-DailyTalk (not available offline): special tokens follow
    src/train_pt.py:104-128 (32000 <|continue|>, 32001 <|correspond|>, 32002+i <|unit i|>), text is byte-level."""
    model_max_length = 400
    pat = re.compile(r"<\|unit(\d+)\|>|<\|correspond\|>|<\|continue\|>|.", re.S)

    def __call__(self, text):
        ids = [1]
        for m in self.pat.finditer(text):
            s = m.group(0)
            if m.group(1) is not None:
                ids.append(32002 + int(m.group(1)))
            elif s == "<|correspond|>":
                ids.append(32001)
            elif s == "<|continue|>":
                ids.append(32000)
            else:
                ids.extend(3 + b for b in s.encode("utf-8"))
        return type("Enc", (), {"input_ids": ids})()

    def decode(self, ids):
        out = []
        for i in (ids.tolist() if torch.is_tensor(ids) else ids):
            if i >= 32002:
                out.append(f"<|unit{i - 32002}|>")
            elif i == 32001:
                out.append("<|correspond|>")
            elif i == 32000:
                out.append("<|continue|>")
            elif 3 <= i < 259:
                out.append(bytes([i - 3]).decode("latin-1"))
        return "".join(out)


def _vb_cfg():
    from tests.golden.configs import SMALL_VB
    return dict(SMALL_VB, n_tokens=10000)   # the LLM may emit any of the 10 000 unit ids


def _models(dev):
    VB_CFG = _vb_cfg()
    from oracle import bigvgan_oracle as BO, voicebox_oracle as VO, w2v_oracle as WO
    from tests.golden.configs import SMALL_VB
    from usdm_amd.llm import USDMForCausalLM
    from usdm_amd.unit_extractor import UnitExtractor
    from usdm_amd.voicebox.model import Voicebox
    from usdm_amd.voicebox.vocoder.env import AttrDict
    from usdm_amd.voicebox.vocoder.models import BigVGAN
    wcfg = dict(WO.XLSR_1B, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_conv_pos_embedding_groups=4, n_units=400)
    ue = UnitExtractor(None, None, device=dev, config=wcfg, state_dict=WO.random_state_dict(wcfg, 1, n_layers=35),
                       centroids=torch.randn(400, 256, generator=torch.Generator().manual_seed(2)))
    lcfg = dict(vocab_size=42003, hidden_size=512, intermediate_size=1024, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=32768)
    llm = USDMForCausalLM.random_init(lcfg, dev, seed=3, ctx_max=512)
    kw = {k: VB_CFG[k] for k in VB_CFG if k != "sigma_min"}
    vb = Voicebox(**kw, attention_dropout=0.0, activation_dropout=0.1, hidden_dropout=0.0, solver="euler", sigma_min=1e-4)
    vb.load_state_dict(VO.random_state_dict(VB_CFG, 4))
    h = AttrDict(dict(BO.BIGVGAN_22K_80, upsample_initial_channel=64))
    voc = BigVGAN(h)
    voc.remove_weight_norm()
    voc.load_state_dict(BO.random_state_dict(h, 5), strict=False)
    return ue, llm, vb.to(dev).eval(), voc.to(dev).eval()


def test_sample_end_to_end_with_reference_prompt(dev, tmp_path):
    from scipy.io.wavfile import read, write
    import usdm_amd.inference as inf
    ue, llm, vb, voc = _models(dev)
    g = torch.Generator().manual_seed(7)
    t = torch.arange(24000) / 16000.0
    wav = (0.2 * torch.sin(2 * torch.pi * 300 * t) + 0.02 * torch.randn(24000, generator=g)).numpy().astype(np.float32)
    user, ref, out = (os.path.join(tmp_path, n) for n in ("user.wav", "ref.wav", "out.wav"))
    write(user, 16000, wav)
    write(ref, 22050, wav[:20000])
    inf.device = dev
    audio = inf.sample(user, ref, llm, ue, vb, voc, ToyTokenizer(), out, n_timesteps=4)
    sr, data = read(out)
    assert sr == 22050 and data.dtype == np.float32 and data.shape == audio.shape
    assert audio.ndim == 1 and audio.size % 256 == 0 and np.isfinite(audio).all() and np.abs(audio).max() <= 1.0
    # the TTS round may only emit unit tokens (ids 32002..42002) or the EOS 28705 (inference.py:53,80-82)
    assert audio.size > 0


class _MelTap:
    """Wraps vocoder.forward to record the mel the Voicebox stage handed over (reconstruct_speech returns audio only)."""

    def __init__(self, voc):
        self.voc, self.orig, self.mel = voc, voc.forward, None

    def __enter__(self):
        def fwd(mel, *a, **k):
            self.mel, self.args = mel.detach().clone().cpu(), a
            return self.orig(mel, *a, **k)
        self.voc.forward = fwd
        return self

    def __exit__(self, *exc):
        self.voc.forward = self.orig


def _snr(x, ref):
    return 10 * np.log10((ref ** 2).sum() / ((x - ref) ** 2).sum())


def _check_vs_oracle(audio, tap, mel_ref, S):
    """The stated tolerances: final mel rel L2 <= 3e-2 vs the fp32 oracle; waveform SNR >= 30 dB vs the fp32
    oracle vocoder ON THE SAME MEL (the vocoder's own error, not the amplified mel difference)."""
    from oracle import bigvgan_oracle as BO
    from usdm_amd.voicebox.util.model_util import mel_mean, mel_std
    assert tap.args == (mel_std, mel_mean)          # de-normalisation folded into the vocoder's layout kernel
    mel = tap.mel                                   # normalised mel [1, 80, S] produced by the HIP Voicebox
    assert mel.shape == mel_ref.shape == (1, 80, S)
    rel = ((mel - mel_ref).norm() / mel_ref.norm()).item()
    h = dict(BO.BIGVGAN_22K_80, upsample_initial_channel=64)
    bsd = BO.random_state_dict(h, 5)
    same_mel = BO.bigvgan_forward(bsd, h, mel * mel_std + mel_mean)[0, 0].clamp(-1, 1).numpy()
    composed = BO.bigvgan_forward(bsd, h, mel_ref * mel_std + mel_mean)[0, 0].clamp(-1, 1).numpy()
    assert audio.shape == same_mel.shape == (256 * S,)
    snr_same, snr_comp = _snr(audio, same_mel), _snr(audio, composed)
    print(f"mel rel L2 {rel:.4f}; waveform SNR vs oracle vocoder on the same mel {snr_same:.1f} dB; vs full oracle composition {snr_comp:.1f} dB")
    assert rel <= 3e-2
    assert snr_same >= 30.0
    return rel, snr_same, snr_comp


def test_reconstruct_speech_without_prompt_matches_oracle_composition(dev):
    from oracle import units_oracle as UO, voicebox_oracle as VO
    from usdm_amd.voicebox.util.model_util import reconstruct_speech
    _, _, vb, voc = _models(dev)
    units = torch.randint(0, 400, (23,), generator=torch.Generator().manual_seed(9))
    frames, _ = UO.process_unit(units.tolist())
    S = len(frames)
    nt = 3
    noise = torch.randn(1, 1, 80, S, generator=torch.Generator().manual_seed(10))
    with _MelTap(voc) as tap:
        audio = reconstruct_speech(units.to(dev), dev, None, None, vb, voc, n_timesteps=nt, noise=noise)
    # oracle composition of the same path (model_util.py:96-104)
    sd = VO.random_state_dict(_vb_cfg(), 4)
    mel = VO.generate(sd, _vb_cfg(), torch.tensor([frames]), torch.zeros(1, 80, S), torch.tensor([S]), nt, [noise[0]], "heun", 1.0, False)
    _check_vs_oracle(audio, tap, mel, S)


def test_reconstruct_speech_with_reference_wav_matches_oracle_composition(dev, tmp_path):
    """The speech-prompt branch (model_util.py:76-95) from a reference WAV FILE: 16 kHz tokenizer units -> process_unit ->
    get_mel (resample 16k -> 22.05k, STFT, mel) -> normalise -> Heun + CFG + prompt re-noising -> slice -> vocoder, against the
    same composition of the CPU oracles."""
    from scipy.io.wavfile import write
    from oracle import mel_oracle as MELO, units_oracle as UO, voicebox_oracle as VO, w2v_oracle as WO
    from usdm_amd.voicebox.util.model_util import mel_mean, mel_std, reconstruct_speech
    ue, _, vb, voc = _models(dev)
    g = torch.Generator().manual_seed(17)
    t = torch.arange(20000) / 16000.0
    wav = (0.2 * torch.sin(2 * torch.pi * 220 * t) + 0.1 * torch.sin(2 * torch.pi * 1330 * t) + 0.02 * torch.randn(20000, generator=g)).float()
    ref_path = os.path.join(tmp_path, "ref16k.wav")
    write(ref_path, 16000, wav.numpy())             # 16 kHz float wav: no host resampling before the tokenizer
    agent = torch.randint(0, 400, (19,), generator=g)
    nt = 4
    # ---- oracle composition
    wcfg = dict(WO.XLSR_1B, hidden_size=256, num_attention_heads=4, intermediate_size=512, num_conv_pos_embedding_groups=4, n_units=400)
    wsd = WO.random_state_dict(wcfg, 1, n_layers=35)
    cen = torch.randn(400, 256, generator=torch.Generator().manual_seed(2))
    ref_units, dist = WO.kmeans_assign(WO.features(wsd, wcfg, wav, 34), cen)
    got_units = ue.predict(wav.to(dev), 34).cpu()
    top2 = torch.topk(dist, 2, largest=False).values
    bad = got_units != ref_units
    assert bool(((top2[:, 1] - top2[:, 0])[bad] <= 1e-4 * dist.abs().max()).all())   # id mismatches only below fp32 noise
    ref_units = got_units                            # (identical unless a centroid near-tie; keep both sides on the same ids)
    r_fr, new_len = UO.process_unit(ref_units.tolist())
    a_fr, _ = UO.process_unit(agent.tolist())
    P, Sa = len(r_fr), len(a_fr)
    S = P + Sa
    mel_p = MELO.get_mel(wav, 16000, new_len)        # [1, 80, P]
    assert mel_p.shape[-1] == P
    cond = torch.zeros(1, 80, S)
    cond[:, :, :P] = (mel_p - mel_mean) / mel_std
    noise = torch.randn(VO.noise_count(nt, "heun", True), 1, 80, S, generator=g)
    sd = VO.random_state_dict(_vb_cfg(), 4)
    unit = torch.tensor([r_fr + a_fr])
    mel = VO.generate(sd, _vb_cfg(), unit, cond, torch.tensor([S]), nt, list(noise), "heun", 1.0, True, torch.tensor([P]))[:, :, P:]
    # ---- HIP path through the reference's call shape (reference_path + token_extractor)
    with _MelTap(voc) as tap:
        audio = reconstruct_speech(agent.to(dev), dev, ref_path, ue, vb, voc, n_timesteps=nt, noise=noise)
    _check_vs_oracle(audio, tap, mel, Sa)
