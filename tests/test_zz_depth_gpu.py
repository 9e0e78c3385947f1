"""GPU parity at the FULL size AND full step count of the BASELINE configs (round-2 verdict, "parity depth"): named test_zz_* so
that these long oracle runs come last in the suite.

  config 1: BigVGAN, the seed-0 [1, 80, 100] mel of SURVEY.md 8d, full 1536-channel generator vs oracle/bigvgan_oracle.py (>= 50 dB)
  config 3: all 32 Mistral-7B layers, 512-token prompt, the FULL 256 generated tokens vs the bf16 CPU oracle (near-tie rule)
  config 4: full-width Voicebox, 500 + 149 units -> S = 1117, speech prompt 256 frames, CFG gs = 1, n_timesteps = 64 = 63 chained NFEs
            (Heun) vs oracle/voicebox_oracle.py; the bf16-operand plan AND the exact-f32 plan side by side ("always
            also report the fp32-kernel variant so bf16 is a measured trade"); final mel rel L2 <= 3e-2, per-NFE trace printed
The CPU oracle side of config 4 takes a few minutes: progress goes to gpurun_out/progress_depth.txt (a silent GPU box reads as hung)."""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Progress(list):
    """A trace list that leaves a heartbeat file behind on every append."""

    def __init__(self, tag):
        super().__init__()
        self.tag, self.t0 = tag, time.time()
        d = os.path.join(ROOT, "gpurun_out")
        self.path = os.path.join(d, "progress_depth.txt") if os.path.isdir(d) else None

    def append(self, x):
        super().append(x)
        if self.path:
            try:
                with open(self.path, "a") as f:
                    f.write(f"{self.tag}: {len(self)} after {time.time() - self.t0:.0f} s\n")
            except OSError:
                pass


def test_config1_bigvgan_seed0_100_frames_full_width_vs_oracle(dev):
    """BASELINE config 1's own workload: mel = N(0,1) * 2.1575 - 5.5419, [1, 80, 100], seed 0 -> 25 600 samples.
    reference: vocoder/models.py:189-211."""
    from oracle import bigvgan_oracle as BO
    from usdm_amd.voicebox.vocoder.env import AttrDict
    from usdm_amd.voicebox.vocoder.models import BigVGAN
    h = AttrDict(BO.BIGVGAN_22K_80)
    sd = BO.random_state_dict(h, 0)
    mel = torch.randn(1, 80, 100, generator=torch.Generator().manual_seed(0)) * 2.1575 - 5.5419
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.time()
    ref = BO.bigvgan_forward(sd, h, mel)
    t_or = time.time() - t0
    voc = BigVGAN(h)
    voc.remove_weight_norm()
    voc.load_state_dict(sd, strict=False)
    wav = voc.to(dev).eval()(mel.to(dev)).cpu()
    assert wav.shape == ref.shape == (1, 1, 25600)
    snr = 10 * torch.log10(ref.pow(2).sum() / (wav - ref).pow(2).sum()).item()
    print(f"config 1 (full-width BigVGAN, 100 frames, seed 0): waveform SNR vs oracle {snr:.1f} dB; CPU oracle {t_or:.2f} s")
    assert snr >= 50.0


def test_config3_llm_all_32_layers_512_prompt_256_tokens_vs_oracle(dev):
    from oracle import mistral_oracle as MO
    from tests._greedy_compare import compare_greedy
    from usdm_amd import synth
    from usdm_amd.inference import generate_bad_words_ids
    from usdm_amd.llm import USDMForCausalLM
    cfg = dict(MO.MISTRAL_7B_USDM)
    L0, new = 512, 256
    sd_dev = synth.random_llm_state_dict(cfg, dev, seed=63)
    m = USDMForCausalLM.from_state_dict(sd_dev, cfg, dev, ctx_max=768)
    sd = {k: v.cpu() for k, v in sd_dev.items()}
    del sd_dev
    torch.cuda.empty_cache()
    ids = torch.randint(32002, 42002, (L0,), generator=torch.Generator().manual_seed(64))
    bad = generate_bad_words_ids(0, 32002, exclude=[28705])
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.time()
    ref, ref_logits = MO.greedy_generate(sd, cfg, ids, new, bad_words_ids=bad, return_logits=True)
    t_or = time.time() - t0
    div, n = compare_greedy(m, dev, ids, ref, ref_logits, new, max_restarts=24, do_sample=True, top_k=1, top_p=1.0, temperature=1.0,
                            bad_words_ids=bad)
    assert n == new and all(32002 <= t < 42003 or t == 28705 for t in ref[L0:])
    print(f"config 3 (32 layers, prompt {L0}, {new} tokens = the whole config): near-tie divergences at {div}; CPU oracle {t_or:.0f} s")
    del m, sd
    torch.cuda.empty_cache()


def test_config4_voicebox_63_nfe_bf16_and_f32_plans_vs_oracle(dev):
    """reference: model/voicebox.py:101-150 (solve_heun, generate), :51-72 (CFG); SURVEY.md 8d config 4."""
    from oracle import voicebox_oracle as VO
    from usdm_amd.voicebox.model import Voicebox
    g = torch.Generator().manual_seed(4)
    S, P, nt, gs = 1117, 256, 64, 1.0                      # 500 agent units -> 861 frames, 149 reference units -> 256 frames
    unit = torch.randint(0, 10000, (1, S), generator=g)
    cond = torch.zeros(1, 80, S)
    cond[:, :, :P] = torch.randn(1, 80, P, generator=g)
    cfg = VO.VOICEBOX_CFG
    sd = VO.random_state_dict(cfg, seed=4)
    noise = [torch.randn(1, 80, S, generator=torch.Generator().manual_seed(5 + i)) for i in range(VO.noise_count(nt, "heun", True))]
    assert len(noise) == 64
    vb = Voicebox(**{k: cfg[k] for k in cfg if k != "sigma_min"}, attention_dropout=0.0, activation_dropout=0.1, hidden_dropout=0.0,
                  solver="euler", sigma_min=1e-4)
    vb.load_state_dict(sd)
    vb = vb.to(dev).eval()
    outs, traces, ms = {}, {}, {}
    for name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        vb.estimator.set_compute_dtype(dt)
        kw = dict(n_timesteps=nt, solver="heun", gradient_scale=gs, speech_prompt=True, prompt_lengths=torch.tensor([P]).to(dev),
                  noise=torch.stack(noise))
        vb.generate(unit.to(dev), cond.to(dev), torch.tensor([S]).to(dev), **kw)       # plan build + graph capture
        torch.cuda.synchronize()
        tr = []
        t0 = time.time()
        outs[name] = vb.generate(unit.to(dev), cond.to(dev), torch.tensor([S]).to(dev), trace=tr, **kw).cpu()
        ms[name] = (time.time() - t0) * 1e3
        traces[name] = [t.cpu() for t in tr]
        assert len(tr) == 63
    vb.estimator.set_compute_dtype(torch.bfloat16)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    tr_ref = _Progress("config 4 oracle NFE")
    t0 = time.time()
    ref = VO.generate(sd, cfg, unit, cond, torch.tensor([S]), nt, noise, "heun", gs, True, torch.tensor([P]), trace=tr_ref)
    t_or = time.time() - t0
    assert len(tr_ref) == 63
    rel, per = {}, {}
    for name in outs:
        per[name] = []
        for raw, vref in zip(traces[name], tr_ref):
            vu, vc = raw[:1], raw[1:]
            per[name].append((((vc + gs * (vc - vu)) - vref).norm() / vref.norm()).item())
        rel[name] = ((outs[name] - ref).norm() / ref.norm()).item()
        # the generated part only (the prompt region is re-noised data the solver overwrites: voicebox.py:115-117)
        gen = ((outs[name][:, :, P:] - ref[:, :, P:]).norm() / ref[:, :, P:].norm()).item()
        show = [f"{per[name][i]:.2e}" for i in (0, 1, 8, 16, 24, 32, 40, 48, 56, 62)]
        print(f"config 4 voicebox, 63 chained NFE (Heun 64, CFG 1.0, prompt 256, S=1117), {name} plan: final mel rel L2 {rel[name]:.3e} "
              f"(generated frames only {gen:.3e}); velocity rel L2 at NFE 1,2,9,17,25,33,41,49,57,63 {show}; max {max(per[name]):.2e}; "
              f"{ms[name]:.0f} ms on the GPU")
    print(f"CPU oracle: {t_or:.0f} s for 63 NFE")
    assert rel["bf16"] <= 3e-2 and per["bf16"][0] <= 1e-2 and max(per["bf16"]) <= 3e-2      # stated tolerance, SURVEY.md 8d
    assert rel["f32"] <= 1e-4 and max(per["f32"]) <= 1e-4                                   # f32 kernels: 1e-4 relative
