"""Average per launch of ssv_diag_kernel of every counter in the rocprofv3 --pmc CSVs under a directory."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
print("pass,counter,launches,avg_per_launch")
for path in sorted(glob.glob(os.path.join(root, "*counter_collection.csv"))):
    name = os.path.basename(path).replace("_counter_collection.csv", "")
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # counter -> dispatch -> value
    with open(path) as f:
        for row in csv.DictReader(f):
            if not ("ssv_diag_kernel" in row["Kernel_Name"] or "ssv_resident_kernel" in row["Kernel_Name"]):
                continue
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for counter in sorted(per):
        vals = list(per[counter].values())
        print(f"{name},{counter},{len(vals)},{sum(vals) / len(vals):.1f}")
