"""Aggregate decode tokens/s of generate_batch on the random-init 7B: python tools/batch_rate.py [B] [new_tokens]."""
import sys, time
import torch
import os as _os; sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from usdm_amd import synth
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
llm = synth.make_llm(dev, ctx_max=2048)
if len(sys.argv) > 3:      # A/B of the decode attention at B > 4: workgroup target of the context split, fused merge
    llm.batch_attn_wgs = int(sys.argv[3])
    llm.batch_fused_merge = len(sys.argv) > 4 and sys.argv[4] == "1"
gen = torch.Generator().manual_seed(3)
prompts = [torch.randint(32002, 42002, (1, 600 - 7 * b), generator=gen).to(dev) for b in range(B)]
llm.generate_batch(prompts, max_new_tokens=24)
torch.cuda.synchronize()
best = 0
for rep in range(2):
    t = time.perf_counter(); llm.generate_batch(prompts, max_new_tokens=8); torch.cuda.synchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter(); llm.generate_batch(prompts, max_new_tokens=8 + n); torch.cuda.synchronize(); t2 = time.perf_counter() - t
    best = max(best, B * n / (t2 - t1))
print(f"B={B} {sys.argv[3:]}: {best:.1f} tok/s aggregate ({1e3 * B / best:.3f} ms per step)")
