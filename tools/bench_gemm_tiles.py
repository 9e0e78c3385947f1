import os, sys, subprocess
names = ["128x128/4w", "128x64/4w", "64x64/4w", "128x128/8w", "128x128/4w/dma", "64x64/4w/dma", "128x64/4w/dma"]
for t in ([int(x) for x in sys.argv[1:]] or range(len(names))):
    print("== tile", names[t], flush=True)
    env = dict(os.environ, USDM_GEMM_TILE=str(t))
    subprocess.run([sys.executable, "tools/bench_gemm.py"], env=env)
