import sys, torch
sys.path.insert(0, '.')
from usdm_amd import ops
dev = torch.device('cuda:0')
M, N, K = 16, 16, 32
for name, A, W in [
    ("ones", torch.ones(M, K), torch.ones(N, K)),
    ("A=k", torch.arange(K).float().repeat(M, 1), torch.ones(N, K)),
    ("W=k", torch.ones(M, K), torch.arange(K).float().repeat(N, 1)),
    ("A=m", torch.arange(M).float()[:, None].repeat(1, K), torch.ones(N, K)),
]:
    out = torch.zeros(M, N, device=dev)
    ops.gemm(A.to(dev), W.to(dev), M=M, N=N, Kc=K, out32=out)
    ref = A @ W.T
    print(name, "out row0:", out[0, :4].tolist(), "ref:", ref[0, :4].tolist(), "out col0:", out[:4, 0].tolist())
# one-hot k probes
for k0 in [0, 1, 4, 5, 16, 17]:
    A = torch.zeros(M, K); A[:, k0] = 1
    W = torch.zeros(N, K); 
    for k1 in range(K): W[:, k1] = k1 + 1
    out = torch.zeros(M, N, device=dev)
    ops.gemm(A.to(dev), W.to(dev), M=M, N=N, Kc=K, out32=out)
    print("A onehot k", k0, "-> picks W k =", out[0, 0].item() - 1)
