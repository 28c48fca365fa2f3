"""Per-shape A/B of the split-precision implicit GEMM (k_conv_limb) against the f32-MFMA direct kernels (k_conv_fast / k_conv_fast_grp) on
the stride-2 convolutions of a ResNet-18 at 640x192 (layerN.0.conv1 3x3, downsample 1x1), forward and data gradient through the C ABI with
cached weight layouts; stand-alone kernel times (hipGraph replay), error of both against float64.
    python scripts/limb_s2_ab.py [batch ...]        -> profiles/round6_limb_s2_ab.log"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
from fusiondepth_amd._lib import call, ptr, stream
from limb_ab import timed

R50_SHAPES = [(128, 128, 48, 160, 3), (256, 256, 24, 80, 3), (512, 512, 12, 40, 3), (256, 512, 48, 160, 1), (512, 1024, 24, 80, 1), (1024, 2048, 12, 40, 1)]
SHAPES = [(64, 128, 48, 160, 3), (64, 128, 48, 160, 1), (128, 256, 24, 80, 3), (128, 256, 24, 80, 1), (256, 512, 12, 40, 3), (256, 512, 12, 40, 1)]


def run(B, ci, co, h, w, k, limb):
    tuning.set_lib(limb_conv=limb)
    g = torch.Generator(device="cuda").manual_seed(ci * 7 + co)
    pad = k // 2
    ho, wo = (h + 2 * pad - k) // 2 + 1, (w + 2 * pad - k) // 2 + 1
    x = torch.randn(B, ci, h, w, device="cuda", generator=g).relu_()
    wt = torch.randn(co, ci, k, k, device="cuda", generator=g) * (2.0 / (ci * k * k)) ** 0.5
    gy = torch.randn(B, co, ho, wo, device="cuda", generator=g)
    plan = FD._conv_plan(x, wt, 2, pad, 0, 0, False)
    dp = plan.dp
    y = torch.empty(B, co, ho, wo, device="cuda"); gx = torch.empty_like(x); gw = torch.empty_like(wt)
    w_ws = torch.empty(plan.weight_ws(), device="cuda")
    f_ws = torch.empty(max(plan.fwd_ws, 1), device="cuda"); f_wt = torch.empty(max(plan.fwd_wt, 1), device="cuda")
    d_ws_n, d_wt_n = plan.data_sizes()
    d_ws = torch.empty(max(d_ws_n, 1), device="cuda"); d_wt = torch.empty(max(d_wt_n, 1), device="cuda")
    call("fd_conv2d_fwd", dp, ptr(x), ptr(wt), None, ptr(y), ptr(f_wt), 0, ptr(f_ws), stream())
    call("fd_conv2d_bwd_data", dp, ptr(gy), ptr(wt), ptr(gx), ptr(d_wt), 0, ptr(d_ws), stream())
    torch.cuda.synchronize()
    t_f = timed(lambda: call("fd_conv2d_fwd", dp, ptr(x), ptr(wt), None, ptr(y), ptr(f_wt), 1, ptr(f_ws), stream()))
    t_d = timed(lambda: call("fd_conv2d_bwd_data", dp, ptr(gy), ptr(wt), ptr(gx), ptr(d_wt), 1, ptr(d_ws), stream()))
    t_w = timed(lambda: call("fd_conv2d_bwd_weight", dp, ptr(x), ptr(gy), ptr(gw), None, ptr(w_ws), 0, stream()))
    return (t_f, t_d, t_w), (x, wt, gy, y, gx, pad, gw)


def main():
    out = open(os.path.join(ROOT, "profiles", "round6_limb_s2_ab_r50.log" if "r50" in sys.argv[1:] else "round6_limb_s2_ab.log"), "w")
    def say(s):
        print(s, flush=True); out.write(s + "\n"); out.flush()
    say("stride-2 convolutions of ResNet-18 @640x192: f32-MFMA direct kernels vs split-precision implicit GEMM (k_conv_limb); us per call alone on "
        "the GPU (incl. split-K finish), TFLOP/s of algorithmic flops, max|err|/max|ref| vs float64")
    args = [a for a in sys.argv[1:] if a != "r50"]
    shapes = R50_SHAPES if "r50" in sys.argv[1:] else SHAPES
    if "r50" in sys.argv[1:]:
        say("(ResNet-50's stride-2 layers: conv2 of layerN.0 and the downsample branch)")
    for B in ([int(a) for a in args] or [12, 24]):
        tot = {0: [0.0, 0.0, 0.0], 1: [0.0, 0.0, 0.0]}
        for (ci, co, h, w, k) in shapes:
            res = {}
            for limb in (0, 1):
                t, (x, wt, gy, y, gx, pad, gw) = run(B, ci, co, h, w, k, limb)
                xd = x.double().requires_grad_(True); wd = wt.double().requires_grad_(True)
                ry = torch.nn.functional.conv2d(xd, wd, None, stride=2, padding=pad)
                rgx, rgw = torch.autograd.grad(ry, (xd, wd), gy.double())
                e = lambda a, r: float((a.double() - r.detach()).abs().max() / r.detach().abs().max())
                res[limb] = (t, e(y, ry), e(gx, rgx), e(gw, rgw))
                for q in range(3):
                    tot[limb][q] += t[q]
            flops = 2.0 * B * ry.shape[2] * ry.shape[3] * ci * co * k * k
            say("b%-2d %4d->%-4d @%3dx%-3d k%d | fwd %6.1f -> %6.1f us (%5.1f -> %5.1f TF) err %.1e -> %.1e | dgrad %6.1f -> %6.1f us (%5.1f -> %5.1f TF) err %.1e -> %.1e"
                % (B, ci, co, h, w, k, res[0][0][0], res[1][0][0], flops / res[0][0][0] / 1e6, flops / res[1][0][0] / 1e6, res[0][1], res[1][1],
                   res[0][0][1], res[1][0][1], flops / res[0][0][1] / 1e6, flops / res[1][0][1] / 1e6, res[0][2], res[1][2]))
            say("%29s| wgrad %6.1f -> %6.1f us (%5.1f -> %5.1f TF) err %.1e -> %.1e"
                % ("", res[0][0][2], res[1][0][2], flops / res[0][0][2] / 1e6, flops / res[1][0][2] / 1e6, res[0][3], res[1][3]))
        say("b%-2d sums: fwd %.1f -> %.1f us, dgrad %.1f -> %.1f us, wgrad %.1f -> %.1f us" % (B, tot[0][0], tot[1][0], tot[0][1], tot[1][1], tot[0][2], tot[1][2]))
    tuning.set_lib(limb_conv=1)


if __name__ == "__main__":
    main()
