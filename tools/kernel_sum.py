import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
def pick(pat): return [(r["Name"][:60], int(r["Calls"]), float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3) for r in rows if pat in r["Name"]]
print(sys.argv[1], "total ms", round(tot/1e6, 2))
for pat in ("k_rmsnorm_fwd", "k_rmsnorm_bwd", "CUDAFunctor_add<c10::BFloat16>", "k_colsum"):
    for n, c, t, a in pick(pat): print("  %-62s %5d calls %8.2f ms  %7.1f us" % (n, c, t, a))
