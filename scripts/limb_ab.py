"""Per-shape A/B of the split-precision (3 x bf16 limbs) GEMM kernels against the f32-MFMA direct kernels on the 1x1 convolutions of a
ResNet-50 at 640x192: forward, data gradient, weight gradient through the C ABI (no autograd in the timed region), stand-alone times
with HIP events, error of both against float64 (max |err| / max |ref| and relative L2).
    python scripts/limb_ab.py [batch ...]           -> profiles/round6_limb_ab.log"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
from fusiondepth_amd._lib import call, ptr, query, stream

SHAPES = [  # Cin, Cout, H, W
    (64, 64, 48, 160), (64, 256, 48, 160), (256, 64, 48, 160), (256, 128, 48, 160),
    (128, 512, 24, 80), (512, 128, 24, 80), (512, 256, 24, 80),
    (256, 1024, 12, 40), (1024, 256, 12, 40), (1024, 512, 12, 40),
    (512, 2048, 6, 20), (2048, 512, 6, 20)]


def timed(fn, n=20, reps=5):
    """kernel time only: n launches captured into a hipGraph, replayed reps times (issued from Python one call costs 10 - 25 us of
    host time - more than most of these kernels take, which is what the first versions of this script measured)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / (n * reps)


def err(a, ref):
    a = a.double()
    return float((a - ref).abs().max() / ref.abs().max()), float((a - ref).norm() / ref.norm())


def run(B, ci, co, h, w, limb, **extra):
    """-> times (fwd, dgrad, wgrad) and outputs, through the C ABI with cached weight layouts"""
    tuning.set_lib(limb_1x1=limb, **extra)
    g = torch.Generator(device="cuda").manual_seed(ci * 7 + co)
    x = torch.randn(B, ci, h, w, device="cuda", generator=g).relu_()          # post-ReLU activations like the network's
    wt = torch.randn(co, ci, 1, 1, device="cuda", generator=g) * (2.0 / ci) ** 0.5
    gy = torch.randn(B, co, h, w, device="cuda", generator=g)
    plan = FD._conv_plan(x, wt, 1, 0, 0, 0, False)
    dp = plan.dp
    y = torch.empty(B, co, h, w, device="cuda"); gx = torch.empty_like(x); gw = torch.empty_like(wt)
    f_ws = torch.empty(max(plan.fwd_ws, 1), device="cuda"); f_wt = torch.empty(max(plan.fwd_wt, 1), device="cuda")
    d_ws_n, d_wt_n = plan.data_sizes()
    d_ws = torch.empty(max(d_ws_n, 1), device="cuda"); d_wt = torch.empty(max(d_wt_n, 1), device="cuda")
    w_ws = torch.empty(plan.weight_ws(), device="cuda")
    call("fd_conv2d_fwd", dp, ptr(x), ptr(wt), None, ptr(y), ptr(f_wt), 0, ptr(f_ws), stream())          # writes the layouts
    call("fd_conv2d_bwd_data", dp, ptr(gy), ptr(wt), ptr(gx), ptr(d_wt), 0, ptr(d_ws), stream())
    torch.cuda.synchronize()
    t_f = timed(lambda: call("fd_conv2d_fwd", dp, ptr(x), ptr(wt), None, ptr(y), ptr(f_wt), 1, ptr(f_ws), stream()))
    t_d = timed(lambda: call("fd_conv2d_bwd_data", dp, ptr(gy), ptr(wt), ptr(gx), ptr(d_wt), 1, ptr(d_ws), stream()))
    t_w = timed(lambda: call("fd_conv2d_bwd_weight", dp, ptr(x), ptr(gy), ptr(gw), None, ptr(w_ws), 0, stream()))
    return (t_f, t_d, t_w), (x, wt, gy, y, gx, gw)


def main():
    out = open(os.path.join(ROOT, "profiles", "round6_limb_ab.log"), "w")
    def say(s):
        print(s, flush=True); out.write(s + "\n"); out.flush()
    say("1x1 stride-1 convolutions of ResNet-50 @640x192: f32-MFMA direct kernels (k_conv_fast / k_wgrad_fast) vs split-precision limb GEMMs "
        "(k_gemm_limb / k_wgrad_limb); us per launch alone on the GPU, TFLOP/s of algorithmic flops, error vs float64 as max|err|/max|ref| (rel. L2)")
    for B in ([int(a) for a in sys.argv[1:]] or [8]):
        tot = {0: [0.0] * 3, 1: [0.0] * 3}
        for (ci, co, h, w) in SHAPES:
            flops = 2.0 * B * h * w * ci * co
            res = {}
            for limb in (0, 1):
                t, (x, wt, gy, y, gx, gw) = run(B, ci, co, h, w, limb)
                w2 = wt.double().view(co, ci)
                ref_y = torch.einsum("oc,nchw->nohw", w2, x.double())
                ref_gx = torch.einsum("oc,nohw->nchw", w2, gy.double())
                ref_gw = torch.einsum("nohw,nchw->oc", gy.double(), x.double()).view(co, ci, 1, 1)
                res[limb] = (t, err(y, ref_y), err(gx, ref_gx), err(gw, ref_gw))
                for k in range(3):
                    tot[limb][k] += t[k]
            line = "b%-2d %4d -> %4d @%2dx%3d %5.2f GF |" % (B, ci, co, h, w, flops / 1e9)
            for k, name in enumerate(("fwd", "dgrad", "wgrad")):
                t0, t1 = res[0][0][k], res[1][0][k]
                e0, e1 = res[0][1 + k], res[1][1 + k]
                line += " %s %6.1f -> %6.1f us (%5.1f -> %5.1f TF, x%.2f) err %.1e (%.1e) -> %.1e (%.1e) |" % (
                    name, t0, t1, flops / t0 / 1e6, flops / t1 / 1e6, t0 / t1, e0[0], e0[1], e1[0], e1[1])
            say(line)
            if os.environ.get("LIMB_SWEEP"):
                for dep in (2, 4):
                    t, _ = run(B, ci, co, h, w, 1, limb_depth=dep)
                    say("      depth %d: fwd %6.1f dgrad %6.1f wgrad %6.1f us" % (dep, t[0], t[1], t[2]))
                tuning.set_lib(limb_depth=2)
        say("b%-2d sum: fwd %.0f -> %.0f us, dgrad %.0f -> %.0f us, wgrad %.0f -> %.0f us" % (
            B, tot[0][0], tot[1][0], tot[0][1], tot[1][1], tot[0][2], tot[1][2]))
    tuning.set_lib(limb_1x1=1)


if __name__ == "__main__":
    main()
