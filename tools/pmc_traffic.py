#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).
usage: tools/pmc_traffic.py fetch.csv write.csv [--min-workgroups N] [--json out.json]
--min-workgroups N keeps only dispatches with at least N workgroups (e.g. the launches of a batch handle).
Counter values are reported as the tool gives them (KB per dispatch); see MI355X_MICROARCH.md for the gfx950 caveats."""
import csv
import json
import re
import sys

args = sys.argv[1:]
min_wg, out_json = 0, None
if "--min-workgroups" in args:
    i = args.index("--min-workgroups"); min_wg = int(args[i + 1]); del args[i:i + 2]
if "--json" in args:
    i = args.index("--json"); out_json = args[i + 1]; del args[i:i + 2]


def load(path, counter):
    agg = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            if int(r["Grid_Size"]) < min_wg * int(r["Workgroup_Size"]):
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"])
            agg.setdefault(name, []).append(float(r["Counter_Value"]))
    return agg


f, w = load(args[0], "FETCH_SIZE"), load(args[1], "WRITE_SIZE")
res = {}
print("| kernel | dispatches | FETCH_SIZE KB/launch (mean / max) | WRITE_SIZE KB/launch (mean / max) |")
print("|---|---|---|---|")
for k in sorted(f, key=lambda k: -sum(f[k])):
    fv, wv = f[k], w.get(k, [0.0])
    res[k] = {"dispatches": len(fv), "fetch_kb_mean": sum(fv) / len(fv), "fetch_kb_max": max(fv),
              "write_kb_mean": sum(wv) / len(wv), "write_kb_max": max(wv)}
    print("| %s | %d | %.1f / %.1f | %.1f / %.1f |" % (k, len(fv), res[k]["fetch_kb_mean"], res[k]["fetch_kb_max"], res[k]["write_kb_mean"], res[k]["write_kb_max"]))
if out_json:
    json.dump(res, open(out_json, "w"), indent=1)
