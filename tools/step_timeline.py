"""Print the kernel timeline of the last bench steps from a rocprofv3 kernel trace: start (us, relative), duration,
idle gap since the previous kernel ended, short name.

usage: python3 tools/step_timeline.py <dir with *kernel_trace.csv> [number of kernels from the end]"""
import csv
import glob
import sys


def main() -> None:
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[-count:]
    t0, prev_end = rows[0][0], rows[0][0]
    busy = 0
    for s, e, name in rows:
        short = name.split("(")[0].replace("void ", "")[-70:]
        print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {short}")
        busy += e - s
        prev_end = max(prev_end, e)
    print(f"span {(prev_end - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
    # the largest idle gaps, with the kernels on either side
    gaps, end = [], rows[0][1]
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        end = max(end, e0)
        gaps.append((s1 - end, n0.split("(")[0][-50:], n1.split("(")[0][-50:], (s1 - t0) / 1e3))
    total = sum(g[0] for g in gaps if g[0] > 0)
    print(f"idle in gaps {total / 1e3:.1f} us; gaps over 15 us:")
    for g in sorted(gaps, reverse=True)[:25]:
        if g[0] > 15000:
            print(f"  {g[0] / 1e3:8.1f} us at {g[3]:10.1f}: {g[1]}  ->  {g[2]}")


if __name__ == "__main__":
    main()
