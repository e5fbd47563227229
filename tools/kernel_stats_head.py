"""First lines of a rocprofv3 kernel-stats CSV found under a directory: average us, calls, name.

    python3 tools/kernel_stats_head.py <dir> [lines]
"""
import csv, glob, sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for r in list(csv.DictReader(open(f)))[:n]:
    print("%9.1f us x %4s  %5.1f%%  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["Percentage"]), r["Name"][:80]))
