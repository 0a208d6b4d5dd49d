#!/usr/bin/env python3
"""Mean counter values per kernel from rocprofv3 --pmc csv output(s).
usage: tools/pmc_table.py counter_collection.csv [more.csv ...] [--min-workgroups N]
Prints one markdown row per kernel with every counter found (mean per dispatch)."""
import csv
import re
import sys

args = sys.argv[1:]
min_wg = 0
if "--min-workgroups" in args:
    i = args.index("--min-workgroups"); min_wg = int(args[i + 1]); del args[i:i + 2]
agg, names = {}, []
for path in args:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if int(r["Grid_Size"]) < min_wg * int(r["Workgroup_Size"]):
                continue
            k = re.sub(r"\(.*", "", r["Kernel_Name"])
            c = r["Counter_Name"]
            if c not in names:
                names.append(c)
            agg.setdefault(k, {}).setdefault(c, []).append(float(r["Counter_Value"]))
print("| kernel | dispatches | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get(names[0], [0]))):
    n = max(len(v) for v in d.values())
    print("| %s | %d | %s |" % (k, n, " | ".join("%.4g" % (sum(d[c]) / len(d[c])) if c in d else "" for c in names)))
