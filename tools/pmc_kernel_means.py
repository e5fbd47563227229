"""Mean counter values per launch of the kernels whose name contains a pattern, from a rocprofv3 counter CSV.

    python3 tools/pmc_kernel_means.py <counter_collection.csv> [pattern]
"""
import collections
import csv
import sys

path, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "lm_gemm")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
csv.field_size_limit(1 << 30)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"]
    if pat not in k:
        continue
    k = k[:48]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    launches[k].add(r["Dispatch_Id"])
for k, v in sorted(acc.items()):
    n = len(launches[k])
    print(k, f"({n} launches)", {c: round(x / n) for c, x in sorted(v.items())})
