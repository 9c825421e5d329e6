"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

FETCH_SIZE / WRITE_SIZE count kilobytes.  On gfx950 FETCH_SIZE tallies 128-byte requests at
64 bytes (MI355X_MICROARCH.md, "HBM"), so it is doubled; the doubling was checked on this
path's own access pattern: bgr2lab_kernel reads exactly 3 B/px with the same 12-byte-per-lane
loads and reports half of that.

usage: python3 tools/pmc_traffic.py <fetch_dir> <write_dir> <npix_per_launch> <key> [out.json]
       key e.g. 1920x1080_K400; merges into profiles/pmc_traffic.json by default"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def mean_per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            name = name.split("(")[0].split("<")[0]
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch_dir, write_dir, npix, key = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(root, "profiles", "pmc_traffic.json")
    fe, wr = mean_per_kernel(fetch_dir, "FETCH_SIZE"), mean_per_kernel(write_dir, "WRITE_SIZE")
    table = json.load(open(out)) if os.path.exists(out) else {}
    entry = table.setdefault(key, {})
    for k in sorted(set(fe) & set(wr)):
        if k.startswith("at::") or "rocclr" in k:
            continue
        rd, wb = 2.0 * fe[k] * 1024.0, wr[k] * 1024.0
        entry[k] = {"read_bytes": rd, "write_bytes": wb, "npix_per_launch": npix,
                    "bytes_per_px": (rd + wb) / npix,
                    "how": "2 x FETCH_SIZE KiB + WRITE_SIZE KiB, mean over launches"}
        print("%-28s read %8.1f MB  write %8.1f MB  %.2f B/px" % (k, rd / 1e6, wb / 1e6, (rd + wb) / npix))
    json.dump(table, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
