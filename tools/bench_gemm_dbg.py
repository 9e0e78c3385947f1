import os, sys, subprocess
for tile in (2, 0):
    for dbg in (0, 1, 2, 4, 8, 3, 6, 7, 15):
        env = dict(os.environ, USDM_GEMM_TILE=str(tile), USDM_GEMM_DBG=str(dbg))
        r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.'); import bench_gemm as b; b.bench(2236,3072,1024); b.bench(4096,4096,4096)"], env=env, capture_output=True, text=True)
        print(f"tile{tile} dbg{dbg:2d}:", " | ".join(l.split(":")[1].strip() for l in r.stdout.strip().splitlines() if "TF/s" in l), flush=True)
