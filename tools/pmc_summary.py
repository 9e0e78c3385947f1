"""Sums a rocprofv3 --pmc counter over the dispatches of kernels matching a substring."""
import csv, sys, glob, collections
path = glob.glob(sys.argv[1])[0]
pat = sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"]:
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in tot:
    print(f"{k}: dispatches={n[k]} sum={tot[k]:.6g} per_dispatch={tot[k]/n[k]:.6g}")
