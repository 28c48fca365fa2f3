#!/usr/bin/env python
"""The kernels of ONE steady-state optimiser step on the busiest HIP stream (the step's serial chain), in launch order, with their
durations: rocprof_stream_chain.py results.db [delimiter=k_adam_dev] [rank of the stream by kernel time = 0 | substring of a kernel name:
the stream that kernel runs on, e.g. k_photo_ms for the depth network's stream]"""
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
sel = sys.argv[3] if len(sys.argv) > 3 else "0"
if sel.lstrip("-").isdigit():
    main = per.most_common(int(sel) + 1)[int(sel)][0]
else:
    main = collections.Counter(st for n, s, e, st in step if sel in n).most_common(1)[0][0]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return re.sub(r"^void ", "", n)[:40]
print("stream %s: %.2f ms of kernels in %d launches (step wall under the profiler %.2f ms)\n" % (main, per[main] / 1e6, sum(1 for r in step if r[3] == main), (hi - lo) / 1e6))
t0 = lo
phase = collections.OrderedDict()
acc = []
prev = None
for n, s, e, st in step:
    if st != main:
        continue
    acc.append((short(n), (e - s) / 1e3, (s - lo) / 1e6, 0.0 if prev is None else max(0, s - prev) / 1e3))
    prev = e
# compress runs; last column: idle time of this stream in front of the kernels of the run (launch latency, or waiting for another stream)
out = []
for n, d, t, gap in acc:
    if out and out[-1][0] == n:
        out[-1][1] += d; out[-1][2] += 1; out[-1][4] += gap
    else:
        out.append([n, d, 1, t, gap])
for n, d, c, t, gap in out:
    print("%8.2f ms  %-40s x%-3d %8.1f us   idle before %7.1f us" % (t, n, c, d, gap))
gaps = sorted((g for _, _, _, g in acc), reverse=True)
print("\nidle on this stream: %.2f ms in total; %d gaps > 20 us hold %.2f ms, the %d gaps <= 20 us %.2f ms (median %.1f us)" % (
    sum(gaps) / 1e3, sum(1 for g in gaps if g > 20), sum(g for g in gaps if g > 20) / 1e3, sum(1 for g in gaps if g <= 20),
    sum(g for g in gaps if g <= 20) / 1e3, gaps[len(gaps) // 2]))

tot = collections.Counter(); cnt = collections.Counter()
for n, d, t, gap in acc:
    tot[n] += d; cnt[n] += 1
print("\nby kernel on this stream:")
for n, d in tot.most_common(40):
    print("  %-42s x%-3d %8.1f us" % (n, cnt[n], d))
