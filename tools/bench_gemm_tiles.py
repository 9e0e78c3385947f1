import os, sys, subprocess
names = {0: "128x128 reg", 1: "128x64 reg", 2: "64x64 reg", 3: "128x128/8w reg", 4: "128x128 dma2", 5: "64x64 dma2", 6: "128x64 dma2",
         7: "64x64 dma3", 8: "64x64 dma4", 9: "128x128 dma4x1ch", 10: "128x64 dma3", 11: "128x128 dma3"}
for t in ([int(x) for x in sys.argv[1:]] or sorted(names)):
    print("== tile", t, names[t], flush=True)
    env = dict(os.environ, USDM_GEMM_TILE=str(t))
    subprocess.run([sys.executable, "tools/bench_gemm.py"], env=env)
