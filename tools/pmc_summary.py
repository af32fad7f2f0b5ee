#!/usr/bin/env python
"""Average rocprofv3 --pmc counter values per launch for kernels whose name contains a substring."""
import csv
import glob
import os
import sys
from collections import defaultdict

root, needle = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for path in sorted(glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    with open(path) as f:
        for row in csv.DictReader(f):
            if needle in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(f"# per-launch averages over kernels matching '{needle}'")
for k, v in sorted(acc.items()):
    print(f"{k:40s} n={len(v):3d}  mean={sum(v)/len(v):.6g}  min={min(v):.6g}  max={max(v):.6g}")
