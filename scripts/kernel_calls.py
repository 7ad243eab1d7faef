"""Per-launch durations of selected kernels from a rocprofv3 kernel trace (kernel_trace.csv): name filter, grid size, duration.
  python scripts/kernel_calls.py <dir with *kernel_trace.csv> k_msm_tblw k_reduce_sets ..."""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
pats = sys.argv[2:]
rows = []
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("cpx::", "").replace("void ", "")
    if any(p in name for p in pats):
        rows.append((int(r["Start_Timestamp"]), name, int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
for t, name, grid, ms in rows:
    print("%-28s grid %9d  %8.3f ms" % (name, grid, ms))
