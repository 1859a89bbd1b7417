"""print (kernel, calls, average ms, total ms) from a rocprofv3 --stats kernel_stats.csv (developer tool)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    if pat in r["Name"]:
        print("%-60s %5s  avg %9.3f ms  total %9.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
