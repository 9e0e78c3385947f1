import os, sys, subprocess
for t in (0, 1, 2):
    print("== tile", ["128x128", "128x64", "64x64"][t], flush=True)
    env = dict(os.environ, USDM_GEMM_TILE=str(t))
    subprocess.run([sys.executable, "tools/bench_gemm.py"], env=env)
