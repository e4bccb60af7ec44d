"""Sums rocprofv3 --pmc counter_collection CSVs per kernel and counter (mean per dispatch)."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[(k, row["Counter_Name"])][0] += float(row["Counter_Value"])
        acc[(k, row["Counter_Name"])][1] += 1
kernels = sorted({k for k, _ in acc})
for k in kernels:
    if not k.startswith("k_"):
        continue
    print(k)
    for (kk, c), (v, n) in sorted(acc.items()):
        if kk == k:
            print("    %-40s mean/dispatch %16.1f   (%d dispatches)" % (c, v / n, n))
