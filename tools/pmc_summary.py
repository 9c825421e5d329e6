"""Summarises a rocprofv3 --pmc counter_collection CSV: mean per dispatch per kernel."""
import csv
import glob
import sys
from collections import defaultdict

pat = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(pat, recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "at::" in k or "rocclr" in k:
        continue
    print(k[:40], " ".join("%s=%.4g(n=%d)" % (c, sum(v) / len(v), len(v)) for c, v in sorted(acc[k].items())))
