"""Per-kernel average durations from a rocprofv3 kernel-trace csv: python tools/kt_summary.py <csv> [substr]"""
import collections
import csv
import sys

d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if sub in k:
        print(f"{k[:60]:60s} n={len(v):5d} avg={sum(v) / len(v) / 1e3:9.1f} us  total={sum(v) / 1e6:8.2f} ms")
