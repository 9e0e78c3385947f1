"""ms per Voicebox NFE at the config-4 shape (B=2 CFG, 1117 frames, bucketed plan, hipGraph replay)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usdm_amd import synth
dev = torch.device("cuda:0")
vb = synth.make_voicebox(dev)
S = 1117
Sb = vb.estimator.bucket_frames(S)
gp, io = vb.estimator.get_plan(1, Sb, 2, True, dev, True)
io["kv_len"].fill_(S + 1)
io["y"].normal_(); io["cond"].normal_(); io["t"].fill_(0.5)
for _ in range(4):
    gp.run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
e0.record()
for _ in range(n):
    gp.run()
e1.record(); torch.cuda.synchronize()
print(f"NFE {e0.elapsed_time(e1) / n:.3f} ms  ({1.734 / (e0.elapsed_time(e1) / n):.0f} TF/s)  launches {len(gp.plan)}")
