#!/usr/bin/env python
"""Concurrency analysis of a rocprofv3 kernel trace: how much of the last optimiser step's wall time has 0/1/2/3+ kernels
in flight, and which kernels run alone (critical-path suspects).  usage: rocprof_timeline.py results.db [steps]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = list(db.execute("select name, start, end, queue_id from kernels order by start"))
# the step boundaries: k_adam_dev marks the end of every step
adam = [i for i, r in enumerate(rows) if "k_adam" in r[0]]
adam = [i for j, i in enumerate(adam) if j + 1 == len(adam) or adam[j + 1] != i + 1]      # last kernel of each Adam group
lo = adam[-2] + 1 if len(adam) >= 2 else 0
hi = adam[-1] + 1
ev = rows[lo:hi]
t0, t1 = ev[0][1], max(r[2] for r in ev)
print("last step: %d kernels, wall %.2f ms, kernel-time sum %.2f ms" % (len(ev), (t1 - t0) / 1e6, sum(r[2] - r[1] for r in ev) / 1e6))
pts = []
for i, r in enumerate(ev):
    pts.append((r[1], 1, i)); pts.append((r[2], -1, i))
pts.sort()
active = set()
hist = defaultdict(float)
alone = defaultdict(float)
last = t0
for t, d, i in pts:
    dt = t - last
    if dt > 0:
        k = len(active)
        hist[min(k, 5)] += dt
        if k == 1:
            alone[next(iter(active))] += dt
    last = t
    if d > 0: active.add(i)
    else: active.discard(i)
tot = t1 - t0
for k in sorted(hist):
    print("  %s kernels in flight: %6.2f ms (%4.1f %%)" % (str(k) if k < 5 else "5+", hist[k] / 1e6, 100 * hist[k] / tot))
byname = defaultdict(float)
for i, dt in alone.items():
    n = re.sub(r"\(anonymous namespace\)::", "", ev[i][0]); n = re.sub(r"\(.*$", "", n)[:70]
    byname[n] += dt
print("time spent as the ONLY kernel in flight, by kernel:")
for n, dt in sorted(byname.items(), key=lambda kv: -kv[1])[:18]:
    print("  %-72s %6.2f ms" % (n, dt / 1e6))

# coarse timeline: per millisecond of the step, the average number of kernels in flight and the kernels that cover it
print("per-ms timeline (avg kernels in flight | top kernels by covered time):")
nb = int((t1 - t0) / 1e6) + 1
cover = [defaultdict(float) for _ in range(nb)]
for r in ev:
    a, b = r[1], r[2]
    n = re.sub(r"\(anonymous namespace\)::", "", r[0]); n = re.sub(r"^void ", "", n); n = re.sub(r"[<(].*$", "", n)[:18]
    k = int((a - t0) / 1e6)
    while a < b and k < nb:
        edge = t0 + (k + 1) * 1e6
        seg = min(b, edge) - a
        cover[k][n] += seg
        a = edge; k += 1
for k in range(nb):
    tot_k = sum(cover[k].values())
    top = sorted(cover[k].items(), key=lambda kv: -kv[1])[:4]
    print("  %2d ms  %.2f  %s" % (k, tot_k / 1e6, "  ".join("%s %.2f" % (n, v / 1e6) for n, v in top)))
