"""Decode tokens/s of the random-init 7B on one GPU: python tools/decode_rate.py [new_tokens] [prompt_len]."""
import sys, time
import torch
import os as _os; sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from usdm_amd import synth

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L0 = int(sys.argv[2]) if len(sys.argv) > 2 else 600
import os
llm = synth.make_llm(dev, ctx_max=2048)
BAN = [[i] for i in range(32002) if i != 28705] if os.environ.get("DECODE_BAN") == "t2u" else None   # the text->unit mask
ids = torch.randint(32002, 42002, (1, L0), generator=torch.Generator().manual_seed(3)).to(dev)
out = llm.generate(input_ids=ids, max_new_tokens=24, bad_words_ids=BAN)      # builds plans + captures the decode graph
torch.cuda.synchronize()
res = []
for rep in range(2):
    t = time.perf_counter(); o1 = llm.generate(input_ids=ids, max_new_tokens=8, bad_words_ids=BAN); torch.cuda.synchronize(); t1 = time.perf_counter() - t
    t = time.perf_counter(); o2 = llm.generate(input_ids=ids, max_new_tokens=8 + n, bad_words_ids=BAN); torch.cuda.synchronize(); t2 = time.perf_counter() - t
    res.append(n / (t2 - t1))
print(f"decode {max(res):.1f} tok/s ({1e3 / max(res):.3f} ms/token), tokens checksum {int(o2.sum())}", flush=True)
