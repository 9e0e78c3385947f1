"""Average duration per (kernel, grid) from a rocprofv3 kernel-trace csv; optional 2nd csv for a side-by-side."""
import csv, sys, collections, re

def load(path, skip_frac=0.5):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[int(len(rows) * skip_frac):]          # steady state only
    by = collections.OrderedDict()
    for r in rows:
        nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        nm = re.sub(r"\(.*", "", nm).replace("unsigned short", "bf16")
        k = (nm[:60], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Z"]))
        by.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return by

a = load(sys.argv[1])
tot = sum(sum(v) for v in a.values())
print(f"total busy {tot/1e3:.0f} us over the second half of the trace")
for k, v in sorted(a.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print(f"{k[0]:60s} wg={k[1]:6d} z={k[2]:3d} n={len(v):4d} avg {sum(v)/len(v)/1e3:8.2f} us  total {sum(v)/1e3:9.0f} us ({100*sum(v)/tot:4.1f}%)")
