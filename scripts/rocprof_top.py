#!/usr/bin/env python
"""Longest individual dispatches of a rocprofv3 results .db, grouped by (kernel, grid): where the step's time concentrates."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
q = "select name, %s, %s, %s, count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels group by 1,2,3,4 order by 6 desc limit 45" % (
    (gx, gx.replace("x", "y"), gx.replace("x", "z")) if gx else ("0", "0", "0"))
print("columns:", cols)
print("| kernel | grid | calls/step | us/step | avg us |\n|---|---|---:|---:|---:|")
for n, a, b, c, k, tot, avg in db.execute(q):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:60]
    print("| `%s` | %sx%sx%s | %.1f | %.0f | %.1f |" % (n, a, b, c, k / steps, tot / steps, avg))
