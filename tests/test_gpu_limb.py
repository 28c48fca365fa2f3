"""Split-precision (3 x bf16 limbs, six products, fp32 accumulate) GEMM kernels of the 1x1 convolutions (csrc/conv_limb.hip: ResNet-50
bottlenecks, networks/resnet_encoder.py:62-74) against float64 and against the f32-MFMA kernels they replace, through the C ABI."""
import numpy as np
import pytest
import torch

from fusiondepth_amd import functional as FD, tuning
from fusiondepth_amd._lib import call, ptr, stream

pytestmark = pytest.mark.gpu

# (batch, Cin, Cout, H, W): bottleneck shapes, split-K shapes (deep layers), channel tails (96, 160), pixel tails (11x38, 5x7), 1216x352 planes
SHAPES = [(2, 64, 256, 48, 160), (2, 256, 64, 48, 160), (8, 1024, 512, 12, 40), (8, 2048, 512, 6, 20), (3, 96, 160, 11, 38),
          (1, 512, 2048, 5, 7), (2, 128, 96, 24, 80), (8, 64, 64, 48, 160)]


def _run(B, ci, co, h, w, limb, bias, act, add):
    tuning.set_lib(limb_1x1=limb)
    try:
        g = torch.Generator(device="cuda").manual_seed(ci * 7 + co)
        x = torch.randn(B, ci, h, w, device="cuda", generator=g).relu_()
        wt = torch.randn(co, ci, 1, 1, device="cuda", generator=g) * (2.0 / ci) ** 0.5
        gy = torch.randn(B, co, h, w, device="cuda", generator=g)
        bs = torch.randn(co, device="cuda", generator=g) if bias else None
        ga = torch.randn(B, ci, h, w, device="cuda", generator=g) if add else None
        gw0 = torch.randn(co, ci, 1, 1, device="cuda", generator=g)
        plan = FD._conv_plan(x, wt, 1, 0, 0, act, False)
        dp = plan.dp
        y = torch.empty(B, co, h, w, device="cuda"); gx = torch.empty_like(x); gw = gw0.clone()
        f_ws = torch.empty(max(plan.fwd_ws, 1), device="cuda"); f_wt = torch.empty(max(plan.fwd_wt, 1), device="cuda")
        d_ws_n, d_wt_n = plan.data_sizes()
        d_ws = torch.empty(max(d_ws_n, 1), device="cuda"); d_wt = torch.empty(max(d_wt_n, 1), device="cuda")
        w_ws = torch.empty(plan.weight_ws(), device="cuda")
        st = stream()
        call("fd_conv2d_fwd", dp, ptr(x), ptr(wt), ptr(bs), ptr(y), ptr(f_wt), 0, ptr(f_ws), st)
        if add:
            call("fd_conv2d_bwd_data_add", dp, ptr(gy), ptr(wt), ptr(ga), ptr(gx), ptr(d_wt), 0, ptr(d_ws), st)
        else:
            call("fd_conv2d_bwd_data", dp, ptr(gy), ptr(wt), ptr(gx), ptr(d_wt), 0, ptr(d_ws), st)
        call("fd_conv2d_bwd_weight", dp, ptr(x), ptr(gy), ptr(gw), None, ptr(w_ws), 1, st)          # accumulate into gw0
        torch.cuda.synchronize()
        return x, wt, gy, bs, ga, gw0, y, gx, gw
    finally:
        tuning.set_lib(limb_1x1=1)


@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "b%d_%d-%d_%dx%d" % s)
@pytest.mark.parametrize("bias,act,add", [(False, 0, False), (True, 1, True)], ids=["plain", "bias-relu-add"])
def test_limb_1x1_keeps_fp32_accuracy(shape, bias, act, add):
    """forward (+ bias, ReLU), data gradient (+ the second gradient joined in the epilogue), weight gradient (accumulated onto an existing
    buffer): error against float64 relative to the scale of the result - 3e-6 (an fp32 dot product's own rounding), and no worse
    than 2x the f32-MFMA kernel's + 1e-7 on the same inputs."""
    import conftest
    B, ci, co, h, w = shape
    res = {}
    for limb in (0, 1):
        x, wt, gy, bs, ga, gw0, y, gx, gw = _run(B, ci, co, h, w, limb, bias, act, add)
        w2 = wt.double().view(co, ci)
        ry = torch.einsum("oc,nchw->nohw", w2, x.double())
        if bias:
            ry = ry + bs.double().view(1, -1, 1, 1)
        if act:
            ry = ry.relu()
        rgx = torch.einsum("oc,nohw->nchw", w2, gy.double())
        if add:
            rgx = rgx + ga.double()
        rgw = torch.einsum("nohw,nchw->oc", gy.double(), x.double()).view(co, ci, 1, 1) + gw0.double()
        e = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
        res[limb] = (e(y, ry), e(gx, rgx), e(gw, rgw))
    for k, name in enumerate(("forward", "data gradient", "weight gradient")):
        bound = max(3e-6, 2 * res[0][k] + 1e-7)
        conftest.report("limb 1x1 %s %s: max |err| / max |ref| vs float64" % ("b%d %d->%d @%dx%d" % shape, name), res[1][k], bound,
                        "(f32-MFMA kernel %.1e)" % res[0][k])
        assert res[1][k] <= bound, "%s: limb %.3g, f32 kernel %.3g" % (name, res[1][k], res[0][k])


def test_limb_route_is_taken_and_logged(capfd):
    """the shapes above really run on the limb kernels (fd_tuning.log names the kernel family per call)"""
    tuning.set_lib(log=1)
    try:
        _run(2, 256, 64, 48, 160, 1, False, 0, False)
    finally:
        tuning.set_lib(log=0)
    err = capfd.readouterr().err
    assert err.count("limb 1x1") >= 3, err


def test_limb_weight_layout_batched_equals_standalone():
    """the pre-split weight image written by the batched re-layout launch (fd_relayout_batch modes 7 / 8) == the stand-alone split
    kernel's, bit for bit, forward and transposed (data gradient)"""
    import ctypes
    from fusiondepth_amd._lib import RelayoutJob, query
    co, ci = 160, 96
    x = torch.randn(2, ci, 8, 16, device="cuda")
    wt = torch.randn(co, ci, 1, 1, device="cuda")
    plan = FD._conv_plan(x, wt, 1, 0, 0, 0, False)
    d_ws_n, d_wt_n = plan.data_sizes()
    for kind, n in ((0, plan.fwd_wt), (1, d_wt_n)):
        assert n == (3 * co * ci + 1) // 2 + (-((3 * co * ci + 1) // 2)) % 4
        a = torch.zeros(n, device="cuda"); b = torch.zeros(n, device="cuda")
        y = torch.empty(2, co, 8, 16, device="cuda"); gx = torch.empty_like(x)
        ws = torch.empty(max(plan.fwd_ws, d_ws_n, 1), device="cuda")
        if kind == 0:
            call("fd_conv2d_fwd", plan.dp, ptr(x), ptr(wt), None, ptr(y), ptr(a), 0, ptr(ws), stream())
        else:
            call("fd_conv2d_bwd_data", plan.dp, ptr(y), ptr(wt), ptr(gx), ptr(a), 0, ptr(ws), stream())
        jobs = (RelayoutJob * 4)()
        nj = query("fd_conv2d_relayout_jobs", plan.dp, kind, ptr(wt), ptr(b), ctypes.addressof(jobs))
        assert nj == 1 and jobs[0].mode == (7 if kind == 0 else 8)
        blocks = query("fd_relayout_plan", ctypes.addressof(jobs), nj)
        dev = torch.frombuffer(bytearray(bytes(memoryview(jobs))[: nj * ctypes.sizeof(RelayoutJob)]), dtype=torch.uint8).cuda()
        call("fd_relayout_batch", ptr(dev), nj, blocks, stream())
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "kind %d" % kind
