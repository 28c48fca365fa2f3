#!/usr/bin/env python
"""The kernels of ONE steady-state optimiser step on the busiest HIP stream (the step's serial chain), in launch order, with their
durations: rocprof_stream_chain.py results.db [delimiter=k_adam_dev]"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
delim = sys.argv[2] if len(sys.argv) > 2 else "k_adam_dev"
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
marks = [e for n, s, e, st in rows if delim in n]
lo, hi = marks[-3], marks[-2]
step = [(n, s, e, st) for n, s, e, st in rows if s >= lo and e <= hi]
per = collections.Counter()
for n, s, e, st in step:
    per[st] += e - s
main = per.most_common(1)[0][0]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return re.sub(r"^void ", "", n)[:40]
print("stream %s: %.2f ms of kernels in %d launches (step wall under the profiler %.2f ms)\n" % (main, per[main] / 1e6, sum(1 for r in step if r[3] == main), (hi - lo) / 1e6))
t0 = lo
phase = collections.OrderedDict()
acc = []
for n, s, e, st in step:
    if st != main:
        continue
    acc.append((short(n), (e - s) / 1e3, (s - lo) / 1e6))
# compress runs
out = []
for n, d, t in acc:
    if out and out[-1][0] == n:
        out[-1][1] += d; out[-1][2] += 1
    else:
        out.append([n, d, 1, t])
for n, d, c, t in out:
    print("%8.2f ms  %-40s x%-3d %8.1f us" % (t, n, c, d))
