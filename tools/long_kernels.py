"""Kernels longer than 1 ms among the last 3200 of a rocprofv3 kernel trace (the LM-head products, the row kernel
and the optimizer of a trainer step), with their durations.

    python3 tools/long_kernels.py <dir with *kernel_trace.csv>
"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for s, e, n in rows[-3200:]:
    if e - s > 1_000_000:
        print(f"{(e - s) / 1e3:9.1f} us  {n[:150]}")
