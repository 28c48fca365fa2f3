#!/usr/bin/env python
"""Concurrency view of a rocprofv3 kernel trace of bench.py: per steady-state optimiser step (between consecutive k_adam_dev
launches) how long the GPU runs 0, 1, 2, 3, 4+ kernels at once, and which kernels fill the time when only ONE is running - the
serial sections of the step (decoder / loss chain) as opposed to the four overlapped encoder passes.
    rocprof_timeline.py results.db [delimiter=k_adam_dev]"""
import collections, re, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
delim = sys.argv[2] if len(sys.argv) > 2 else "k_adam_dev"
rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
marks = [e for n, s, e, st in rows if delim in n]
if len(marks) < 3:
    sys.exit("need at least 3 %s launches" % delim)
lo, hi = marks[1], marks[-1]
steps = len(marks) - 2
ev = []
for n, s, e, st in rows:
    if s >= lo and e <= hi:
        ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort(key=lambda t: (t[0], t[1]))
busy = collections.Counter()
alone = collections.Counter()
active = {}
t_prev, depth = lo, 0
for t, d, n in ev:
    dt = t - t_prev
    if dt > 0:
        busy[min(depth, 4)] += dt
        if depth == 1:
            alone[next(iter(active))] += dt
    if d > 0:
        active[n] = active.get(n, 0) + 1
    else:
        active[n] -= 1
        if active[n] == 0:
            del active[n]
    depth += d
    t_prev = t
busy[min(depth, 4)] += hi - t_prev
wall = (hi - lo) / steps / 1e6
print("# concurrency over %d steady-state steps (%.2f ms per step under the profiler)\n" % (steps, wall))
print("| kernels in flight | ms per step | share |\n|---|---:|---:|")
for k in range(5):
    print("| %s | %.2f | %.1f %% |" % (("%d" % k) if k < 4 else "4+", busy[k] / steps / 1e6, 100.0 * busy[k] / (hi - lo)))
print("\nTime with exactly one kernel in flight, by kernel:\n\n| kernel | ms per step |\n|---|---:|")
for n, dt in alone.most_common(14):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)[:70]
    print("| `%s` | %.2f |" % (n, dt / steps / 1e6))

# ---- per stream: kernel time per step (a lower bound of the step = the busiest stream's chain) and its first / last kernel
per = collections.defaultdict(lambda: [0.0, 0, None, None])
for n, s, e, st in rows:
    if s >= lo and e <= hi:
        q = per[st]
        q[0] += e - s; q[1] += 1
print("\nKernel time per HIP stream (sum of kernel durations, per step):\n\n| stream | ms per step | launches per step | top kernels (ms per step) |\n|---|---:|---:|---|")
for st, q in sorted(per.items(), key=lambda kv: -kv[1][0]):
    top = collections.Counter()
    for n, s, e, st2 in rows:
        if st2 == st and s >= lo and e <= hi:
            nn = re.sub(r"\(anonymous namespace\)::", "", n); nn = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", nn); nn = re.sub(r"^void ", "", nn)[:28]
            top[nn] += e - s
    print("| %s | %.2f | %.0f | %s |" % (st, q[0] / steps / 1e6, q[1] / steps, ", ".join("%s %.2f" % (k, v / steps / 1e6) for k, v in top.most_common(5))))
