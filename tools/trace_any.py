"""Per-kernel totals over the last N kernels of a rocprofv3 kernel trace: python tools/trace_any.py trace.csv N"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
busy = collections.defaultdict(int); cnt = collections.defaultdict(int)
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]
    busy[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[name] += 1
wall = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"{len(rows)} kernels, wall {wall/1e3:.1f} us, busy {sum(busy.values())/1e3:.1f} us")
for k in sorted(busy, key=lambda k: -busy[k]):
    print(f"  {k:50s} n={cnt[k]:4d} total {busy[k]/1e3:9.1f} us  avg {busy[k]/cnt[k]/1e3:7.2f} us")
