import sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_gemm as b
b.bench(4096, 4096, 4096, reps=3)
