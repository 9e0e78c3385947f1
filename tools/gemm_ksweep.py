import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_gemm as b
for K in (64, 128, 256, 512, 1024, 2048, 4096):
    b.bench(2236, 3072, K, reps=50)
for K in (64, 1024):
    b.bench(2236, 3072, K, out16=False, reps=50)
