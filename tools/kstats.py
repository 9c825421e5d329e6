"""Short per-kernel table from a rocprofv3 --kernel-trace --stats csv directory.
usage: python3 tools/kstats.py <dir> [divide_calls_by]"""
import csv, glob, re, sys
d = sys.argv[1]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0.0
for r in rows:
    name = r["Name"]
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0][:60]
    t = float(r["TotalDurationNs"]) / 1e6 / div
    tot += t
    if name.startswith("at::") or "rocclr" in name:      # torch kernels of the input synthesis, not the path
        tot -= t
        continue
    if t > 0.005:
        print("%-62s calls %7.1f  total %9.3f ms  avg %9.1f us" % (name, int(r["Calls"]) / div, t, float(r["AverageNs"]) / 1e3))
print("sum %.2f ms" % tot)
