#!/usr/bin/env python
"""Turn a rocprofv3 results .db (kernel trace) into the per-kernel summary table committed under profiles/."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                       "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary (%s)" % sys.argv[1].split("/")[-1])
print("total kernel time %.2f ms over %g optimiser steps = %.2f ms/step; %d launches (%.0f per step); %d distinct kernels\n"
      % (tot, steps, tot / steps, sum(r[1] for r in rows), sum(r[1] for r in rows) / steps, len(rows)))
print("| kernel | calls | total ms | % | avg us | min us | max us |")
print("|---|---:|---:|---:|---:|---:|---:|")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    n = re.sub(r"\(anonymous namespace\)::", "", r[0])
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:90]
    print("| `%s` | %d | %.2f | %.1f | %.1f | %.1f | %.1f |" % (n, r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5]))
