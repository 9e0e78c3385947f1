"""Asymptotic streaming rate of usdm_gemv (one huge matrix) vs the decode shapes: how much is per-launch ramp."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_gemv as b
for N in (4096, 16384, 65536, 262144):
    b.bench(N, 4096, copies=max(2, (1 << 31) // (N * 8192)), reps=3)
