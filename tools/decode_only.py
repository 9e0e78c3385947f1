"""Runs N decode steps of the random-init 7B on one GPU (for rocprofv3 --pmc passes over the GEMV kernel)."""
import sys
import torch
import os as _os; sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from usdm_amd import synth

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 16
llm = synth.make_llm(dev, ctx_max=1024)
ids = torch.randint(32002, 42002, (1, 512), generator=torch.Generator().manual_seed(3)).to(dev)
out = llm.generate(input_ids=ids, max_new_tokens=steps)
torch.cuda.synchronize()
print("generated", out.shape[1] - 512, "tokens; weight bytes/token", llm.weight_bytes_per_token())
