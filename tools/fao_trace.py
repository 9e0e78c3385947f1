"""Phase stamps of the fused attention + o_proj decode launch (needs USDM_EXTRA_HIPCC_FLAGS=-DUSDM_FAO_TRACE on llm_fused_k.hip)."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import synth, _lib
dev = torch.device("cuda:0")
llm = synth.make_llm(dev, ctx_max=2048)
ids = torch.randint(32002, 42002, (1, 600), generator=torch.Generator().manual_seed(3)).to(dev)
llm.generate(input_ids=ids, max_new_tokens=40)
torch.cuda.synchronize()
buf = np.zeros(256 * 16, dtype=np.uint64)
assert _lib.lib.usdm_dbg_fao_trace(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size)) == 0
t = buf.reshape(256, 16).astype(np.int64)
t0 = t[:, [0, 8]].min()
names = ["start", "loads issued", "partials published", "combine done", "x gathered", "after barrier", "rows summed"]
for role, off in (("wave 0 (attention)", 0), ("wave 4 (loader)", 8)):
    print(role)
    for i, nm in enumerate(names):
        v = (t[:, off + i] - t0) / 100.0
        v = v[t[:, off + i] > 0]
        if len(v):
            print(f"  {nm:22s} min {v.min():6.2f}  median {np.median(v):6.2f}  max {v.max():6.2f} us   (combine workgroups 0..31 median {np.median((t[:32, off + i] - t0) / 100.0):6.2f})")
