"""Summarise rocprofv3 --pmc counter_collection.csv files: mean per launch of the forward kernel."""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for path in glob.glob(f"{d}/*counter_collection.csv"):
        with open(path) as f:
            seen = set()
            for row in csv.DictReader(f):
                if "fwd_kernel" in row["Kernel_Name"]:
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    if row["Dispatch_Id"] not in seen:
                        seen.add(row["Dispatch_Id"])
                        agg["_dur_ns"].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in sorted(agg.items()):
        print(f"{d:24s} {k:36s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
