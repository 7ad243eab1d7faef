#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats CSV (…_kernel_stats.csv) -> markdown table."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---:|---:|---:|---:|---:|---:|")
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")
    print("| %s | %s | %.2f | %.1f | %.1f | %.1f | %.2f |" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                          float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["Percentage"])))
