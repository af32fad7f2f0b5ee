#!/usr/bin/env python
"""print (calls, avg us, total ms, name) rows of a rocprofv3 kernel_stats.csv, longest total first"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in rows[:top]:
    print("%6d %10.1f us %9.3f ms  %s" % (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Name"][:110]))
