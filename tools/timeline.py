#!/usr/bin/env python3
"""Print a window of a rocprofv3 (rocpd sqlite) kernel trace as a timeline: start (us, relative), duration, gap to the previous kernel
of the same queue, queue id, kernel name.   usage: tools/timeline.py results.db [first_kernel_index] [count]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = [c for c in cols if "queue" in c.lower()]
scol = [c for c in cols if "stream" in c.lower()]
sel = qcol[0] if qcol else (scol[0] if scol else "0")
rows = cur.execute("select %s, start, end, %s from kernels order by start" % (name_col, sel)).fetchall()
i0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 120
t0 = rows[i0][1]
last_end = {}
print("cols:", cols)
for name, s, e, q in rows[i0:i0 + n]:
    name = re.sub(r"\(.*", "", name)[:28]
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    print("%9.1f  dur %6.1f  gap %6.1f  q=%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, name))
