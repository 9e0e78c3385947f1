"""Times the non-LLM stages separately (ms): tokenizer 10 s, one Voicebox NFE (B=2, 1117 frames), BigVGAN 861 frames."""
import sys, time
import torch
import os as _os; sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from usdm_amd import synth
dev = torch.device("cuda:0")

def timeit(f, n=5):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e3 / n

ue = synth.make_unit_extractor(dev)
wave = torch.randn(160000, device=dev) * 0.1
print("tokenizer ms", round(timeit(lambda: ue.predict(wave, 34)), 2), flush=True)
del ue
vb = synth.make_voicebox(dev)
S = 1117
x = torch.randint(0, 10000, (2, S), device=dev); y = torch.randn(2, 80, S, device=dev)
t = torch.full((2, 1, 1), 0.5, device=dev); L = torch.tensor([S, S], device=dev)
print("voicebox NFE ms", round(timeit(lambda: vb.estimator(x, y, y, t, L), 10), 3), flush=True)
del vb
voc = synth.make_bigvgan(dev)
mel = torch.randn(1, 80, 861, device=dev) * 2.1575 - 5.5419
print("bigvgan 861 frames ms", round(timeit(lambda: voc(mel)), 2), flush=True)
