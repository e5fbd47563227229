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


if __name__ == "__main__":
    main()
