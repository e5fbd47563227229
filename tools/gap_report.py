"""Where the GPU waits: reads a rocprofv3 kernel trace (CSV) and reports the stretches in which no kernel ran.

    python3 tools/gap_report.py <kernel_trace.csv> [shortest gap in us, default 200] [how many to list, default 25]

Printed: the span from the first kernel's start to the last kernel's end, the time some kernel was running, the
idle time by gap length, and the longest gaps with the kernel that ended before and the one that started after each.
"""
from __future__ import annotations

import csv
import sys


def main() -> None:
    path = sys.argv[1]
    floor_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    span = max(e for _, e, _ in rows) - rows[0][0]
    busy, gaps = 0, []
    reach, last = rows[0][0], ""
    for s, e, name in rows:
        if s > reach:
            gaps.append((s - reach, reach - rows[0][0], last, name))
            busy += e - s
        else:
            busy += max(0, e - reach)
        if e > reach:
            reach, last = e, name
    idle = sum(g[0] for g in gaps)
    print(f"{len(rows)} kernels over {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms, idle {idle / 1e6:.1f} ms "
          f"({100.0 * idle / span:.1f} %)")
    for lo, hi in ((0, 10), (10, 100), (100, 1000), (1000, 10000), (10000, 1e12)):
        sel = [g[0] for g in gaps if lo * 1e3 <= g[0] < hi * 1e3]
        print(f"  gaps of {lo:>5} .. {hi if hi < 1e9 else 'inf':>5} us: {len(sel):7d}, {sum(sel) / 1e6:9.1f} ms")
    big = sorted((g for g in gaps if g[0] >= floor_us * 1e3), reverse=True)[:top]
    print(f"longest gaps (at least {floor_us:.0f} us), in the order they happened:")
    for d, at, before, after in sorted(big, key=lambda g: g[1]):
        print(f"  at {at / 1e6:9.1f} ms  {d / 1e3:9.0f} us  after {before[:60]!r}  before {after[:60]!r}")


if __name__ == "__main__":
    main()
