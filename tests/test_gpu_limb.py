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


# ---- stride-2 convolutions on the split-precision implicit GEMM (k_conv_limb): ResNet layerN.0.conv1 (3x3) and downsample (1x1) -------------
# (batch, Cin, Cout, H, W, K): the ResNet-18 shapes of the step (640x192, stacked batch 12 / 24), a split-K shape, channel / pixel tails
# (1x1 downsample layers go to the limb kernels only where the launch fills the chip without split-K: the small ones here are controls)
S2_SHAPES = [(4, 64, 128, 48, 160, 3), (4, 64, 128, 48, 160, 1), (6, 128, 256, 24, 80, 3), (12, 256, 512, 12, 40, 3), (12, 256, 512, 12, 40, 1),
             (3, 96, 160, 11, 38, 3), (2, 64, 96, 10, 14, 3), (1, 128, 64, 7, 9, 1), (2, 64, 64, 13, 32, 3), (5, 128, 128, 6, 24, 3),
             (8, 256, 512, 48, 160, 1), (4, 512, 1024, 24, 80, 1)]       # ResNet-50's downsample layers: large enough for the limb kernels


def _run_s2(B, ci, co, h, w, k, limb, bias, act, add):
    tuning.set_lib(limb_conv=limb)
    try:
        g = torch.Generator(device="cuda").manual_seed(ci * 5 + co + k)
        x = torch.randn(B, ci, h, w, device="cuda", generator=g).relu_()
        wt = torch.randn(co, ci, k, k, device="cuda", generator=g) * (2.0 / (ci * k * k)) ** 0.5
        pad = k // 2
        ho, wo = (h + 2 * pad - k) // 2 + 1, (w + 2 * pad - k) // 2 + 1
        gy = torch.randn(B, co, ho, wo, device="cuda", generator=g)
        bs = torch.randn(co, device="cuda", generator=g) if bias else None
        ga = torch.randn(B, ci, h, w, device="cuda", generator=g) if add else None
        plan = FD._conv_plan(x, wt, 2, pad, 0, act, False)
        dp = plan.dp
        y = torch.empty(B, co, ho, wo, device="cuda"); gx = torch.empty_like(x)
        gw0 = torch.randn(co, ci, k, k, device="cuda", generator=g)
        gw = gw0.clone() if add else torch.empty_like(gw0)
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
        call("fd_conv2d_bwd_weight", dp, ptr(x), ptr(gy), ptr(gw), None, ptr(w_ws), 1 if add else 0, st)      # add: accumulate onto gw0
        torch.cuda.synchronize()
        return x, wt, gy, bs, ga, y, gx, pad, gw, (gw0 if add else None)
    finally:
        tuning.set_lib(limb_conv=1)


@pytest.mark.parametrize("shape", S2_SHAPES, ids=lambda s: "b%d_%d-%d_%dx%d_k%d" % s)
@pytest.mark.parametrize("bias,act,add", [(False, 0, False), (True, 1, True)], ids=["plain", "bias-relu-add"])
def test_limb_stride2_conv_keeps_fp32_accuracy(shape, bias, act, add):
    """forward (+ bias, ReLU), data gradient (four output-parity classes in one grouped launch, + a second gradient joined in the
    epilogue) and weight gradient (one tap per workgroup; + accumulation onto an existing buffer) of the stride-2 layers: error against float64 relative to the scale of the result <= 3e-6 and no worse than 2x the
    f32-MFMA direct kernel's + 1e-7 on the same inputs."""
    import conftest
    B, ci, co, h, w, k = shape
    res = {}
    for limb in (0, 1):
        x, wt, gy, bs, ga, y, gx, pad, gw, gw0 = _run_s2(B, ci, co, h, w, k, limb, bias, act, add)
        xd = x.double().cpu().requires_grad_(True)
        wd = wt.double().cpu().requires_grad_(True)
        ry = torch.nn.functional.conv2d(xd, wd, bs.double().cpu() if bias else None, stride=2, padding=pad)
        rgx, rgw = torch.autograd.grad(ry, (xd, wd), gy.double().cpu())
        if gw0 is not None:
            rgw = rgw + gw0.double().cpu()
        if act:
            ry = ry.relu()
        if add:
            rgx = rgx + ga.double().cpu()
        e = lambda a, r: float((a.double().cpu() - r.detach()).abs().max() / r.detach().abs().max())
        res[limb] = (e(y, ry), e(gx, rgx), e(gw, rgw))
    for kk, name in enumerate(("forward", "data gradient", "weight gradient")):
        bound = max(3e-6, 2 * res[0][kk] + 1e-7)
        conftest.report("limb stride-2 %s %s: max |err| / max |ref| vs float64" % ("b%d %d->%d @%dx%d k%d" % shape, name), res[1][kk], bound,
                        "(f32-MFMA kernel %.1e)" % res[0][kk])
        assert res[1][kk] <= bound, "%s: limb %.3g, f32 kernel %.3g" % (name, res[1][kk], res[0][kk])


def test_limb_stride2_route_is_taken_and_logged(capfd):
    tuning.set_lib(log=1)
    try:
        _run_s2(4, 64, 128, 48, 160, 3, 1, False, 0, False)
    finally:
        tuning.set_lib(log=0)
    err = capfd.readouterr().err
    assert err.count("limb direct") >= 3, err


@pytest.mark.parametrize("k", [3])
def test_limb_stride2_weight_layout_batched_equals_standalone(k):
    """re-layout modes 9 / 10 (the taps' matrix pre-split into bf16 limbs, forward and the parity classes of the data gradient) written by
    the batched launch == the stand-alone split kernel's image, bit for bit"""
    import ctypes
    from fusiondepth_amd._lib import RelayoutJob, query
    co, ci, pad = 160, 96, k // 2
    x = torch.randn(2, ci, 10, 16, device="cuda")
    wt = torch.randn(co, ci, k, k, device="cuda")
    plan = FD._conv_plan(x, wt, 2, pad, 0, 0, False)
    d_ws_n, d_wt_n = plan.data_sizes()
    ho, wo = (10 + 2 * pad - k) // 2 + 1, (16 + 2 * pad - k) // 2 + 1
    for kind, n in ((0, plan.fwd_wt), (1, d_wt_n)):
        a = torch.zeros(n, device="cuda"); b = torch.zeros(n, device="cuda")
        y = torch.randn(2, co, ho, wo, device="cuda"); gx = torch.empty_like(x)
        ws = torch.empty(max(plan.fwd_ws, d_ws_n, 1), device="cuda")
        if kind == 0:
            call("fd_conv2d_fwd", plan.dp, ptr(x), ptr(wt), None, ptr(y), ptr(a), 0, ptr(ws), stream())
        else:
            call("fd_conv2d_bwd_data", plan.dp, ptr(y), ptr(wt), ptr(gx), ptr(a), 0, ptr(ws), stream())
        jobs = (RelayoutJob * 4)()
        nj = query("fd_conv2d_relayout_jobs", plan.dp, kind, ptr(wt), ptr(b), ctypes.addressof(jobs))
        assert nj == (1 if kind == 0 else 4) and all(jobs[i].mode == (9 if kind == 0 else 10) for i in range(nj))
        blocks = query("fd_relayout_plan", ctypes.addressof(jobs), nj)
        dev = torch.frombuffer(bytearray(bytes(memoryview(jobs))[: nj * ctypes.sizeof(RelayoutJob)]), dtype=torch.uint8).cuda()
        call("fd_relayout_batch", ptr(dev), nj, blocks, stream())
        torch.cuda.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), "kind %d" % kind
