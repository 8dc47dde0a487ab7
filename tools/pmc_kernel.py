"""Per-kernel sums of one rocprofv3 --pmc pass (counter_collection.csv): python tools/pmc_kernel.py <csv> [name filter]"""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-70:]
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[k].add(r["Dispatch_Id"])
for k, c in agg.items():
    print(k, "dispatches", len(n[k]))
    for name, v in sorted(c.items()):
        print(f"   {name:32s} {v / len(n[k]):16.1f} per dispatch")
