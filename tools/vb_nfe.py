"""Runs a few Voicebox estimator evaluations at the config-4 shape (B=2 CFG, 1117 frames) for profiling."""
import sys
import torch
import os as _os; sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from usdm_amd import synth
dev = torch.device("cuda:0")
vb = synth.make_voicebox(dev)
S = 1117
x = torch.randint(0, 10000, (2, S), device=dev)
y = torch.randn(2, 80, S, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    out = vb.estimator(x, y, y, torch.full((2, 1, 1), 0.5, device=dev), torch.tensor([S, S], device=dev))
torch.cuda.synchronize()
print(out.shape)
