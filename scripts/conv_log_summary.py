import sys, re, collections
cnt = collections.Counter(); fl = collections.Counter()
for l in sys.stdin:
    if not l.startswith("FDCONV"): continue
    m = re.match(r"FDCONV (\w+) (.+?) N=(\d+) Cin=(\d+) H=(\d+) W=(\d+) Cout=(\d+) K=(\d+) s=(\d+) pad_mode=(\d+)", l)   # the path may be several words
    if not m: continue
    what, path, N, Ci, H, W, Co, K, s, pm = m.group(1), m.group(2), *map(int, m.groups()[2:])
    Ho, Wo = (H + s - 1) // s, (W + s - 1) // s
    f = 2.0 * N * Ho * Wo * Co * Ci * K * K
    key = (what, path, "%dx%d s%d %d->%d @%dx%d N%d%s" % (K, K, s, Ci, Co, H, W, N, " refl" if pm else ""))
    cnt[key] += 1; fl[key] += f
tot = sum(fl.values())
bypath = collections.Counter()
for k, v in fl.items(): bypath[(k[0], k[1])] += v
for k, v in sorted(bypath.items(), key=lambda kv: -kv[1]): print("%-6s %-8s %6.1f GF %5.1f %%" % (k[0], k[1], v / 1e9, 100 * v / tot))
print("--- largest non-Winograd entries")
for k, v in sorted(fl.items(), key=lambda kv: -kv[1]):
    if not k[1].startswith("wino"): print("%-6s %-8s %-44s x%-3d %6.1f GF %4.1f %%" % (k[0], k[1], k[2], cnt[k], v / 1e9, 100 * v / tot))
