"""Per-kernel durations and inter-kernel gaps of the LAST decode step in a rocprofv3 kernel trace."""
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1])[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last argmax_final and the one before it: one decode step in between
idx = [i for i, r in enumerate(rows) if "argmax_final" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
step = rows[a + 1:b + 1]
t0 = int(rows[a]["End_Timestamp"])
tot = int(step[-1]["End_Timestamp"]) - t0
busy = collections.defaultdict(int); cnt = collections.defaultdict(int)
gap_total = 0
prev_end = t0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:40]
    busy[name] += e - s; cnt[name] += 1
    gap_total += max(0, s - prev_end)
    prev_end = max(prev_end, e)
print(f"step wall {tot/1e3:.1f} us, kernels {len(step)}, sum of gaps {gap_total/1e3:.1f} us ({100*gap_total/tot:.1f}%)")
for k in sorted(busy, key=lambda k: -busy[k]):
    print(f"  {k:42s} n={cnt[k]:3d} total {busy[k]/1e3:8.1f} us  avg {busy[k]/cnt[k]/1e3:6.2f} us")
