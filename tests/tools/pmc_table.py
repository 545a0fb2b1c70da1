"""Aggregate rocprofv3 --pmc CSV passes into one per-kernel table (sums over dispatches)."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.Counter()
dur = collections.defaultdict(float)
for f in sorted(glob.glob(f"{root}/*/p_counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and f.endswith("/1/p_counter_collection.csv"):
            seen.add(key)
            disp[k] += 1
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k in sorted(agg, key=lambda k: -dur[k]):
    if not k.startswith("bg::"):
        continue
    v = agg[k]
    print(f"== {k}  dispatches {disp[k]}  total {dur[k]:.3f} ms (profiled)")
    for name in sorted(v):
        print(f"     {name:32s} {v[name]:18.0f}")
