import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
for s, e, n in rows[-3200:]:
    if e - s > 1_000_000:
        print(f"{(e - s) / 1e3:9.1f} us  {n[:150]}")
