"""Does a line touched once stay in the 256 MB Infinity Cache?  GEMV over a cold matrix vs the same matrix after a
one-dword-per-64-B touch pass (both out of a 1.5 GB ring so nothing is resident by accident)."""
import sys
import torch
sys.path.insert(0, ".")
from usdm_amd import ops
dev = torch.device("cuda:0")
for (N, K) in ((4096, 4096), (6144, 4096), (4096, 14336), (28672, 4096)):
    nbytes = N * K * 2
    copies = max(3, int(1.5e9 // nbytes))
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(copies)]
    x = torch.randn(K, device=dev).to(torch.bfloat16)
    y = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    def run(W):
        ops.gemv(W, x, N=N, K=K, y16=y)
    def touch(W, stride):
        return W.view(torch.int32).view(-1, stride // 4)[:, 0].sum()
    ev = lambda: torch.cuda.Event(enable_timing=True)
    for mode in ("cold", "touch64", "touch128", "hot"):
        ts = []
        for rep in range(3):
            for W in Ws:
                if mode.startswith("touch"):
                    touch(W, int(mode[5:]))
                elif mode == "hot":
                    run(W)
                e0, e1 = ev(), ev()
                e0.record(); run(W); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"N{N} K{K} {nbytes/1e6:.0f} MB {mode:9s}: median {med:7.2f} us  {nbytes/med/1e3:7.1f} GB/s", flush=True)
