#!/usr/bin/env python
"""Turn a rocprofv3 results .db (kernel trace) into the per-kernel summary table committed under profiles/.

    rocprof_summary.py results.db <steps> [rows] [delimiter-kernel]

With a delimiter kernel (e.g. k_adam_dev: exactly one launch per optimiser step) only the STEADY-STATE window is summarised:
everything between the end of the first and the end of the last delimiter launch, i.e. (count - 1) whole steps - the
parameter-flattening copies, allocator fills and autotune launches of start-up are left out of the per-step figures."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
where, note = "", ""
if len(sys.argv) > 4:
    ends = [r[0] for r in db.execute("select end from kernels where name like ? order by end", ("%" + sys.argv[4] + "%",))]
    skip = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # leave out the first `skip` steps as well (the steps in which replay.py records)
    if len(ends) >= skip + 2:
        ends = ends[skip:]
        where = " where start >= %d and end <= %d" % (ends[0], ends[-1])
        steps = len(ends) - 1
        note = " (steady state: between the %s and the last `%s`)" % ("first" if skip == 0 else "%d-th" % (skip + 1), sys.argv[4])
rows = list(db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                       "from kernels" + where + " group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats summary (%s)%s" % (sys.argv[1].split("/")[-1], note))
print("total kernel time %.2f ms over %g optimiser steps = %.2f ms/step; %d launches (%.0f per step); %d distinct kernels\n"
      % (tot, steps, tot / steps, sum(r[1] for r in rows), sum(r[1] for r in rows) / steps, len(rows)))
aten = sum(r[2] for r in rows if "at::native" in r[0] or "rocclr" in r[0])
print("ATen / runtime glue (at::native::*, __amd_rocclr_*): %.2f %% of the kernel time, %.0f launches per step\n"
      % (100 * aten / tot, sum(r[1] for r in rows if "at::native" in r[0] or "rocclr" in r[0]) / steps))
print("| kernel | calls | total ms | % | avg us | min us | max us |")
print("|---|---:|---:|---:|---:|---:|---:|")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    n = re.sub(r"\(anonymous namespace\)::", "", r[0])
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:90]
    print("| `%s` | %d | %.2f | %.1f | %.1f | %.1f | %.1f |" % (n, r[1], r[2], 100 * r[2] / tot, r[3], r[4], r[5]))
