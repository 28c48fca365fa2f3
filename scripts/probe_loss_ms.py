"""Loss-path probe (bench.py's roofline_loss_path workload) for the multi-scale kernel: fd_photo_ms_fwd + fd_photo_ms_bwd over
the four scales, timed from a replayed hipGraph with HIP events; also the per-scale kernels for comparison.

    python scripts/probe_loss_ms.py [B=12] [rows_per_strip=0] [loops=10]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LOOPS = int(sys.argv[3]) if len(sys.argv) > 3 else 10
H, W = 192, 640
batch = synthetic.make_batch(B, H, W, seed=1)
po = FD.PhotoOptions()
tgt = batch[("color", 0, 0)]
srcs = [batch[("color", -1, 0)], batch[("color", 1, 0)]]
ident = torch.empty(B, 2, H, W, device="cuda")
for i, s_ in enumerate(srcs):
    FD.reprojection_loss_map(s_, tgt, True, out=ident[:, i:i + 1])
I = torch.eye(4, device="cuda").repeat(B, 1, 1); I[:, 0, 3] = 0.05
disps = [torch.rand(B, 1, H >> s, W >> s, device="cuda").mul_(0.1).add_(0.02).requires_grad_(True) for s in range(4)]
noise = torch.randn(4, B, 2, H, W, device="cuda")
G = 2 if B % 2 == 0 else 1


def step_ms():
    photo, si, sel = FD.photo_loss_ms(disps, [I, I], batch[("K", 0)], batch[("inv_K", 0)], srcs, tgt, ident, list(noise),
                                      batch["4beam"], (0, 1, 2, 3), po, G, ROWS)
    tot = 0
    for s in range(4):
        tot = tot + photo[s] + si[s]
    tot.backward()
    return tot.detach()


def step_old():
    tot = 0
    for s in range(4):
        photo, si = FD.photo_loss(disps[s], [I, I], batch[("K", 0)], batch[("inv_K", 0)], srcs, tgt, ident, noise[s], batch["4beam"], po, False, G)[:2]
        tot = tot + photo + si
    tot.backward()
    return tot.detach()


def graph_time_us(fn, launches=5, replays=10):
    for d in disps:
        d.grad = None               # AccumulateGrad must first run on the capture stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(launches):
                fn()
        g.replay(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(replays):
            g.replay()
        b.record(); torch.cuda.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    return a.elapsed_time(b) * 1e3 / (launches * replays)


for d in disps: d.grad = None
t_ms = step_ms(); g_ms = [d.grad.clone() for d in disps]
for d in disps: d.grad = None
t_old = step_old(); g_old = [d.grad.clone() for d in disps]
print("total loss ms %.7f old %.7f" % (float(t_ms), float(t_old)))
for s in range(4):
    print("scale %d: grad rel-L1 diff %.3g" % (s, float((g_ms[s] - g_old[s]).abs().sum() / g_old[s].abs().sum())))
if os.environ.get("PROBE_TIMING", "1") == "1":
    which = os.environ.get("PROBE_WHICH", "both")
    us_old = graph_time_us(step_old) if which in ("both", "old") else float("nan")
    print("old timed", flush=True)
    us_ms = graph_time_us(step_ms) if which in ("both", "ms") else float("nan")
    byts = 343.7 * H * W * B
    print("ms  path: %.1f us  -> %.1f GB/s  frac %.3f" % (us_ms, byts / us_ms / 1e3, byts / us_ms / 1e3 / 8000))
    print("old path: %.1f us  -> %.1f GB/s  frac %.3f" % (us_old, byts / us_old / 1e3, byts / us_old / 1e3 / 8000))
for _ in range(LOOPS):
    step_ms()
torch.cuda.synchronize(); print("done")
