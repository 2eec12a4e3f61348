"""Prints the kernel sequence of the last update in a rocprofv3 --kernel-trace csv (updates end with reduce_adam):
python tools/trace_last_update.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "reduce_adam" in r["Kernel_Name"]]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%7.1f %6.1f  %-62s wg %s x %s x %s" % ((s - t0) / 1000, (e - s) / 1000, r["Kernel_Name"][:62],
          int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), r["Grid_Size_Y"], r["Grid_Size_Z"]))
print("launches", b - a, "span_us", (int(rows[b - 1]["End_Timestamp"]) - t0) / 1000)
