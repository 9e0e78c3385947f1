"""BigVGAN forward time at the bench shape (861 mel frames) per MFMA operand mode."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import bigvgan_oracle as BO
from usdm_amd.voicebox.vocoder.env import AttrDict
from usdm_amd.voicebox.vocoder.models import BigVGAN
dev = torch.device("cuda:0")
h = AttrDict(BO.BIGVGAN_22K_80)
sd = BO.random_state_dict(h, 0)
T = int(os.environ.get("BV_T", "861"))
mel = (torch.randn(1, 80, T, generator=torch.Generator().manual_seed(0)) * 2.1575 - 5.5419).to(dev)
modes = os.environ.get("BV_MODES", "f32,bf16,bf16x3").split(",")
outs = {}
for name in modes:
    cd = {"f32": torch.float32, "bf16": torch.bfloat16}.get(name, name)
    try:
        voc = BigVGAN(h, compute_dtype=cd)
    except ValueError as e:
        print(name, "unsupported:", e); continue
    voc.remove_weight_norm(); voc.load_state_dict(sd, strict=False); voc = voc.to(dev).eval()
    for _ in range(3):
        w = voc(mel)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        w = voc(mel)
    torch.cuda.synchronize()
    outs[name] = w.float().cpu()
    print(f"{name}: {(time.time() - t0) * 100:.2f} ms per forward", flush=True)
ref = outs.get("f32")
if ref is not None:
    for k, v in outs.items():
        if k != "f32":
            print(f"{k} vs f32 plan: SNR {10 * torch.log10(ref.pow(2).sum() / (v - ref).pow(2).sum()).item():.1f} dB")
