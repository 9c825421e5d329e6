"""One line of the bench's figures: python tools/show_bench.py <file with the JSON line>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("step %.3f ms  %.0f Mpx/s  path %.4f | dcd %.3f ms (%.3f) | %s" % (
    d["ms_per_step"], d["value"], d.get("path", {}).get("frac", 0), r["launch_ms"], r["frac"],
    "  ".join("%s %.3f" % (k["kernel"].split("_kernel")[0], k["launch_ms"]) for k in d["roofline_kernels"])))
for c in d.get("configs", []):
    cfg = c.get("config", "")
    name = cfg.get("workload", "") if isinstance(cfg, dict) else str(cfg)
    print("   %-44s %.3f ms  %.0f Mpx/s  path %s" % (name[:44], c["ms_per_step"], c["value"],
                                                   c.get("path", {}).get("frac") if isinstance(c.get("path"), dict) else ""))
    for k in c.get("roofline_kernels", []):
        print("        %-50s %.3f ms  %s" % (k["kernel"][:50], k["launch_ms"], k.get("frac")))
