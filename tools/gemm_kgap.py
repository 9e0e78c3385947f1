"""Kernel duration vs launch period for small-K GEMMs (run under rocprofv3 --kernel-trace)."""
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_gemm as b
for K in (64, 1024):
    b.bench(2236, 3072, K, reps=50)
