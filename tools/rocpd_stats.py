#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / median / min / max (us).  (The median is the figure to
quote for a kernel whose first launch carries a one-off of several ms — code object load, LDS re-partition —, which the average does not hide.)
usage: tools/rocpd_stats.py results.db [out.md] [--grid-z N]   (--grid-z: only dispatches whose grid has N workgroups/items in z,
e.g. the launches of a batch handle)"""
import re
import sqlite3
import sys

grid_z = None
if "--grid-z" in sys.argv:
    i = sys.argv.index("--grid-z")
    grid_z = int(sys.argv[i + 1])
    del sys.argv[i:i + 2]
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
where = ""
if grid_z is not None:
    zc = [c for c in cols if "grid" in c.lower() and c.lower().endswith("z")]
    wz = [c for c in cols if "workgroup" in c.lower() and c.lower().endswith("z")]
    if not zc:
        sys.exit("no grid-z column among: %s" % cols)
    # rocprofv3 reports the grid in work-items: z items = z workgroups x workgroup-size z (1 for every kernel here)
    where = " where %s = %d" % (zc[0], grid_z) if not wz else " where %s = %d * %s" % (zc[0], grid_z, wz[0])
rows = cur.execute("select %s, start, end from kernels%s" % (name_col, where)).fetchall()
agg = {}
for name, s, e in rows:
    name = re.sub(r"\(.*", "", name)
    a = agg.setdefault(name, [0, 0.0, 1e30, 0.0, []])
    d = (e - s) / 1e3
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d); a[4].append(d)
tot = sum(a[1] for a in agg.values())
lines = ["| kernel | calls | total us | avg us | median us | min us | max us | % |", "|---|---|---|---|---|---|---|---|"]
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.2f | %.1f |" % (name, a[0], a[1], a[1] / a[0], sorted(a[4])[len(a[4]) // 2], a[2], a[3], 100 * a[1] / tot))
txt = "\n".join(lines)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
