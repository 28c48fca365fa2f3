"""GPU parity tests for the network stack: MFMA implicit-GEMM conv (fwd / dgrad / wgrad), fused BatchNorm,
max-pool, decoder input assembly, Adam, and the HIP-backed ``networks`` modules against the CPU oracle
and the reference-generated golden vectors.  1e-4 relative (north star) unless noted."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import inputs as gin
from conftest import assert_close, check_grad_compact
from oracle import layers as OL
from oracle import networks as ON

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def FD():
    from fusiondepth_amd import functional
    return functional


@pytest.fixture(scope="module")
def NW():
    from fusiondepth_amd import networks
    return networks


def dev(t):
    return t.detach().clone().cuda()


def cpu(t):
    return t.detach().cpu().numpy()


def relclose(got, want, what, rtol=1e-4, arel=1e-4):
    want = np.asarray(want)
    assert_close(got, want, rtol=rtol, atol=arel * max(float(np.abs(want).max()), 1e-30), what=what)


ACTS = {"none": lambda v: v, "relu": F.relu, "elu": F.elu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}

CONV_CASES = [
    # N, Cin, H, W, Cout, K, stride, pad, mode, act, bias, in_norm
    (2, 3, 32, 48, 64, 7, 2, 3, "zero", "none", False, True),      # RGB stem
    (2, 2, 32, 48, 64, 7, 2, 3, "zero", "none", False, True),      # beam stem
    (1, 6, 30, 44, 64, 7, 2, 3, "zero", "none", False, True),      # pose-pair stem, odd output sizes
    (2, 64, 16, 24, 64, 3, 1, 1, "zero", "none", False, False),    # layer1
    (2, 64, 16, 24, 128, 3, 2, 1, "zero", "none", False, False),   # layer2.0.conv1
    (2, 64, 16, 24, 128, 1, 2, 0, "zero", "none", False, False),   # downsample
    (2, 64, 15, 23, 128, 3, 2, 1, "zero", "none", False, False),   # odd input, stride 2
    (2, 64, 15, 23, 128, 1, 2, 0, "zero", "none", False, False),
    (1, 256, 6, 20, 512, 3, 2, 1, "zero", "none", False, False),   # layer4.0.conv1 (full-size shape)
    (2, 64, 8, 12, 256, 1, 1, 0, "zero", "none", False, False),    # bottleneck 1x1
    (2, 512, 2, 3, 256, 3, 1, 1, "reflect", "elu", True, False),   # upconv(4,0) at 64x96 input
    (2, 96, 16, 24, 32, 3, 1, 1, "reflect", "elu", True, False),   # upconv(1,1)
    (1, 16, 64, 96, 16, 3, 1, 1, "reflect", "elu", True, False),   # upconv(0,1)
    (2, 16, 33, 50, 1, 3, 1, 1, "reflect", "sigmoid", True, False),  # dispconv, ragged
    (2, 22, 16, 24, 1, 3, 1, 1, "reflect", "tanh", True, False),   # refiner dispconv
    (2, 5, 16, 24, 7, 3, 1, 1, "zero", "none", True, False),       # Conv3x3(use_refl=False)
    (2, 512, 2, 3, 256, 1, 1, 0, "zero", "relu", True, False),     # pose squeeze
    (2, 256, 2, 3, 256, 3, 1, 1, "zero", "relu", True, False),     # pose conv
    (2, 256, 2, 3, 12, 1, 1, 0, "zero", "none", True, False),      # pose out
    (2, 16, 20, 28, 32, 5, 2, 2, "zero", "relu", True, False),     # PoseCNN 5x5
    (1, 64, 48, 160, 64, 3, 1, 1, "zero", "none", False, False),   # layer1 at full 192x640 resolution
    (2, 3, 32, 48, 64, 7, 2, 3, "zero", "none", False, False),     # stems on a pre-normalised input: dedicated wgrad kernel
    (1, 6, 30, 44, 64, 7, 2, 3, "zero", "none", False, False),
    (2, 2, 37, 131, 64, 7, 2, 3, "zero", "none", False, False),    # ragged tiles, odd sizes
    (3, 4, 64, 160, 64, 7, 2, 3, "zero", "none", False, False),
    (6, 6, 192, 640, 64, 7, 2, 3, "zero", "none", False, False),   # full size: 1 440 tiles on 512 persistent workgroups (the software-pipelined tile loop)
    (2, 32, 37, 130, 16, 3, 1, 1, "reflect", "elu", True, False),  # upconv(0,0): narrow wgrad kernel, 2 channel groups, ragged tiles
    (3, 32, 9, 70, 1, 3, 1, 1, "reflect", "sigmoid", True, False), # dispconv(1) shape class, Cout = 1
    (2, 16, 21, 67, 12, 3, 1, 1, "zero", "none", False, False),    # narrow kernel with zero padding
    (12, 16, 96, 128, 16, 3, 1, 1, "reflect", "none", False, False),  # > 512 tiles: persistent workgroups take several
    (2, 32, 37, 130, 32, 3, 1, 1, "reflect", "elu", True, False),  # Refiner decoder 32 -> 32: narrow wgrad kernel with two row groups
    (2, 32, 21, 67, 24, 3, 1, 1, "zero", "none", False, False),    # ... partial second row group, zero padding
    (6, 32, 96, 320, 32, 3, 1, 1, "reflect", "elu", True, False),  # ... at the Refiner's half-resolution level (persistent loop)
]


@pytest.mark.parametrize("N,Cin,H,W,mode,act", [
    (2, 16, 33, 50, "reflect", "sigmoid"),       # dispconv(0) class, ragged
    (1, 128, 6, 20, "reflect", "sigmoid"),       # dispconv(3)
    (2, 8, 2, 2, "reflect", "none"),             # every pixel is a corner: rows -1 and 2 mirror onto rows 1 and 0
    (1, 4, 3, 3, "reflect", "tanh"),             # H = W = 3: the centre has three pre-images per direction
    (2, 16, 3, 7, "reflect", "sigmoid"),
    (2, 5, 4, 6, "zero", "none"),                # Conv3x3(use_refl=False), odd channel count
    (12, 16, 192, 640, "reflect", "sigmoid"),    # full size
])
def test_conv3x3_single_output_channel_stencils(FD, N, Cin, H, W, mode, act, fdtune):
    """conv_c1.hip: Cout = 1 (dispconv) forward and data gradient as stencils - the data gradient with the adjoint of the reflection
    padding folded into its tap sums - against torch on the CPU, and against the GEMM kernels these layers used before."""
    import ctypes
    from fusiondepth_amd import _lib
    d = _lib.ConvDesc(N, Cin, H, W, 1, 3, 3, 1, 1, 1 if mode == "reflect" else 0, {"none": 0, "relu": 1, "elu": 2, "sigmoid": 3, "tanh": 4}[act], 0)
    assert _lib.query("fd_conv2d_bwd_data_wt_floats", ctypes.byref(d)) == 0
    rng = np.random.RandomState(N * 100 + Cin + H)
    x = torch.from_numpy(rng.randn(N, Cin, H, W).astype(np.float32))
    w = torch.from_numpy((rng.randn(1, Cin, 3, 3) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32))
    b = torch.from_numpy((0.1 * rng.randn(1)).astype(np.float32))
    xo, wo, bo = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = ACTS[act](F.conv2d(F.pad(xo, (1,) * 4, mode="reflect"), wo, bo) if mode == "reflect" else F.conv2d(xo, wo, bo, 1, 1))
    cot = torch.from_numpy(rng.randn(*yo.shape).astype(np.float32))
    want = torch.autograd.grad((yo * cot).sum(), [xo, wo, bo])
    res = {}
    for c1 in ("1", "0"):
        fdtune.lib(conv_c1=int(c1))
        xg, wg, bg = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
        yg = FD.conv2d(xg, wg, bg, 1, 1, mode, act)
        got = torch.autograd.grad((yg * dev(cot)).sum(), [xg, wg, bg])
        relclose(cpu(yg), cpu(yo), "conv fwd (conv_c1=%s)" % c1)
        relclose(cpu(got[0]), cpu(want[0]), "conv dgrad (conv_c1=%s)" % c1)
        relclose(cpu(got[1]), cpu(want[1]), "conv wgrad")
        relclose(cpu(got[2]), cpu(want[2]), "conv bias grad")
        res[c1] = (yg.detach(), got[0])
    relclose(cpu(res["1"][0]), cpu(res["0"][0]), "stencil vs GEMM forward", arel=2e-6)
    relclose(cpu(res["1"][1]), cpu(res["0"][1]), "stencil vs GEMM data gradient", arel=2e-6)


@pytest.mark.parametrize("n16", [1, 0])
@pytest.mark.parametrize("N,Cin,Cout,H,W,mode,act", [
    (1, 16, 16, 64, 96, "reflect", "elu"),       # upconv(0,1) at a small input: one-and-a-half column tiles
    (2, 16, 16, 37, 130, "reflect", "elu"),      # ragged: the first padding column is an interior lane of the last tile, 5 row tiles
    (2, 32, 16, 37, 130, "reflect", "elu"),      # upconv(0,0): 8 channel groups, 4-row tiles
    (2, 16, 32, 21, 67, "reflect", "none"),      # 32 output channels (the data gradient's shape of upconv(0,0))
    (2, 16, 16, 21, 67, "zero", "relu"),         # zero padding
    (3, 16, 16, 2, 2, "reflect", "sigmoid"),     # both mirrors inside one tile
    (12, 16, 16, 192, 640, "reflect", "elu"),    # the real thing: 2 880 workgroups
])
def test_conv3x3_n16_kernel(FD, N, Cin, Cout, H, W, mode, act, n16, fdtune):
    """conv_n16.hip (16 / 32-channel 3x3 blocks on the 16x16x4 MFMA, weights in registers, the patch staged once): forward, data
    gradient (kernel on dY with flipped weights + the ring of the reflect adjoint) and weight gradient against torch on the CPU;
    n16 = 0 runs the same cases on the implicit-GEMM kernel these layers used before."""
    import ctypes
    from fusiondepth_amd import _lib
    fdtune.lib(conv_n16_min_pixels=1 if n16 else -1, reflect_ring=2)      # reflect_ring=2: the interior + ring data gradient on every plane size
    d = _lib.ConvDesc(N, Cin, H, W, Cout, 3, 3, 1, 1, 1 if mode == "reflect" else 0, {"none": 0, "relu": 1, "elu": 2, "sigmoid": 3}[act], 0)
    assert (_lib.query("fd_conv2d_fwd_wt_floats", ctypes.byref(d)) == 0) == bool(n16)      # the n16 kernel reads the weights as they are
    rng = np.random.RandomState(N * 100 + Cin + H)
    x = torch.from_numpy(rng.randn(N, Cin, H, W).astype(np.float32))
    w = torch.from_numpy((rng.randn(Cout, Cin, 3, 3) * np.sqrt(2.0 / (Cin * 9))).astype(np.float32))
    b = torch.from_numpy((0.1 * rng.randn(Cout)).astype(np.float32))
    xo, wo, bo = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yo = ACTS[act](F.conv2d(F.pad(xo, (1,) * 4, mode="reflect"), wo, bo) if mode == "reflect" else F.conv2d(xo, wo, bo, 1, 1))
    cot = torch.from_numpy(rng.randn(*yo.shape).astype(np.float32))
    want = torch.autograd.grad((yo * cot).sum(), [xo, wo, bo])
    xg, wg, bg = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    yg = FD.conv2d(xg, wg, bg, 1, 1, mode, act)
    got = torch.autograd.grad((yg * dev(cot)).sum(), [xg, wg, bg])
    relclose(cpu(yg), cpu(yo), "conv fwd")
    relclose(cpu(got[0]), cpu(want[0]), "conv dgrad")
    relclose(cpu(got[1]), cpu(want[1]), "conv wgrad")
    relclose(cpu(got[2]), cpu(want[2]), "conv bias grad")


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(str(v) for v in c))
def test_conv2d_fwd_bwd(FD, case):
    N, Cin, H, W, Cout, K, stride, pad, mode, act, has_bias, in_norm = case
    import zlib
    rng = np.random.RandomState(zlib.crc32(repr(case).encode()) % (2 ** 31))
    x = torch.from_numpy(rng.rand(N, Cin, H, W).astype(np.float32) if in_norm else rng.randn(N, Cin, H, W).astype(np.float32))
    w = torch.from_numpy((rng.randn(Cout, Cin, K, K) * np.sqrt(2.0 / (Cin * K * K))).astype(np.float32))
    b = torch.from_numpy((0.1 * rng.randn(Cout)).astype(np.float32)) if has_bias else None
    xo, wo = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    bo = b.clone().requires_grad_(True) if has_bias else None
    xin = (xo - 0.45) / 0.225 if in_norm else xo
    if mode == "reflect":
        yo = F.conv2d(F.pad(xin, (pad,) * 4, mode="reflect"), wo, bo, stride)
    else:
        yo = F.conv2d(xin, wo, bo, stride, pad)
    yo = ACTS[act](yo)
    cot = torch.from_numpy(rng.randn(*yo.shape).astype(np.float32))
    want = torch.autograd.grad((yo * cot).sum(), [xo, wo] + ([bo] if has_bias else []))

    xg, wg = dev(x).requires_grad_(True), dev(w).requires_grad_(True)
    bg = dev(b).requires_grad_(True) if has_bias else None
    yg = FD.conv2d(xg, wg, bg, stride, pad, mode, act, in_norm)
    got = torch.autograd.grad((yg * dev(cot)).sum(), [xg, wg] + ([bg] if has_bias else []))
    relclose(cpu(yg), cpu(yo), "conv fwd")
    relclose(cpu(got[0]), cpu(want[0]), "conv dgrad")
    relclose(cpu(got[1]), cpu(want[1]), "conv wgrad")
    if has_bias:
        relclose(cpu(got[2]), cpu(want[2]), "conv bias grad")


@pytest.mark.parametrize("min_cout", [32, 64])
@pytest.mark.parametrize("N,Cin,Cout,H,W,mode", [
    (2, 96, 32, 24, 40, "reflect"),      # upconv(1,1)'s channels: 1.5 input-channel tiles, half an output-channel tile
    (3, 64, 32, 12, 20, "reflect"),      # upconv(1,0)'s
    (2, 64, 48, 8, 12, "zero"),          # zero padding: the data gradient is itself a Winograd convolution with 64 "output" channels
    (2, 96, 32, 7, 10, "reflect"),       # odd height: F(2,3) along x / the 1-D weight gradient
])
def test_conv3x3_with_fewer_than_64_output_channels_on_the_winograd_kernels(FD, fdtune, N, Cin, Cout, H, W, mode, min_cout):
    """fd_tuning.wino_min_cout / wino_wgrad_min_cout = 32 route the decoder's 32-channel blocks (forward, data gradient, weight gradient)
    to the Winograd kernels with rows of their 64-channel tile idle; 64 keeps them on the direct kernels.  Both against torch float64."""
    fdtune.lib(wino_min_cout=min_cout, wino_wgrad_min_cout=min_cout)
    torch.manual_seed(N * 100 + Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, device="cuda", requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    b = (torch.randn(Cout, device="cuda") * 0.1).requires_grad_(True)
    y = FD.conv2d(x, w, b, 1, 1, mode, "elu")
    cot = torch.randn_like(y)
    got = torch.autograd.grad((y * cot).sum(), [x, w, b])
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    xp = F.pad(xd, (1, 1, 1, 1), mode="reflect" if mode == "reflect" else "constant")
    yd = F.elu(F.conv2d(xp, wd, bd))
    want = torch.autograd.grad((yd * cot.double()).sum(), [xd, wd, bd])
    relclose(cpu(y), cpu(yd.float()), "y", rtol=1e-5, arel=3e-6)
    for g, r, what in zip(got, want, ("dx", "dw", "db")):
        relclose(cpu(g), cpu(r.float()), what, rtol=1e-4, arel=1e-5)


@pytest.mark.parametrize("fields", [dict(grp_tile64_below=100000), dict(wino_wgrad_xcd_few=0), dict(wino_fwd_halfm=0, wino_wgrad_halfm=0),
                                    dict(wino_min_cout=64, wino_wgrad_min_cout=64)],
                         ids=lambda f: ",".join("%s=%s" % kv for kv in f.items()))
def test_round5_tuning_switches_keep_the_results(FD, fdtune, fields):
    """The A/B switches fd_tuning grew in round 5 choose between kernels, not between results: a ResNet stage transition (stride-2 3x3 + 1x1
    downsample: the grouped data gradient on 64x64 tiles), a 256-channel layer whose weight gradient has 4 pixel slices (3-D grid instead of
    the XCD-aware one) and a 32-output-channel reflect-padded block (wave-pair variants off / direct kernels) against torch float64."""
    fdtune.lib(wino_fwd_2dp_min_wgs=1, **fields)
    torch.manual_seed(4242)
    cases = [(2, 64, 128, 24, 40, 3, 2, "zero"), (2, 64, 128, 24, 40, 1, 2, "zero"), (1, 256, 256, 12, 40, 3, 1, "zero"),
             (2, 96, 32, 24, 40, 3, 1, "reflect")]
    for N, Ci, Co, H, W, K, stride, mode in cases:
        x = torch.randn(N, Ci, H, W, device="cuda", requires_grad=True)
        w = (torch.randn(Co, Ci, K, K, device="cuda") * 0.05).requires_grad_(True)
        y = FD.conv2d(x, w, None, stride, K // 2, mode, "none")
        cot = torch.randn_like(y)
        gx, gw = torch.autograd.grad((y * cot).sum(), [x, w])
        xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
        xp = F.pad(xd, (K // 2,) * 4, mode="reflect" if mode == "reflect" else "constant")
        yd = F.conv2d(xp, wd, None, stride)
        gxd, gwd = torch.autograd.grad((yd * cot.double()).sum(), [xd, wd])
        what = "%dx%d k%d s%d %s" % (Ci, Co, K, stride, mode)
        relclose(cpu(y), cpu(yd.float()), "y " + what, rtol=1e-5, arel=3e-6)
        relclose(cpu(gx), cpu(gxd.float()), "dx " + what, rtol=1e-4, arel=1e-5)
        relclose(cpu(gw), cpu(gwd.float()), "dw " + what, rtol=1e-4, arel=1e-5)


@pytest.mark.parametrize("fields", [dict(limb_conv=0), dict(wino_wgrad_limb=0), dict(wino_wgrad_limb=1), dict(limb_1x1=0), dict(limb_conv=0, wino_wgrad_limb=0, limb_1x1=0),
                                    dict(wino_fwd_limb=1)],
                         ids=lambda f: ",".join("%s=%s" % kv for kv in f.items()))
def test_round6_tuning_switches_keep_the_results(FD, fdtune, fields):
    """The split-precision routes of round 6 (fd_tuning.limb_conv / wino_wgrad_limb / limb_1x1) choose between kernels, not between
    results: a ResNet stage transition (3x3 stride 2 + 1x1 downsample), a large 1x1 stride-2 layer, 3x3 layers whose weight gradient takes
    the limb matrix loop (zero and reflect padding) and a 1x1 bottleneck layer, all three directions against torch float64 at the bounds
    of the f32 kernels."""
    fdtune.lib(**fields)
    torch.manual_seed(777)
    cases = [(2, 64, 128, 24, 40, 3, 2, "zero"), (2, 64, 128, 24, 40, 1, 2, "zero"), (8, 256, 512, 24, 80, 1, 2, "zero"), (2, 256, 256, 12, 40, 3, 1, "zero"),
             (2, 128, 64, 24, 40, 3, 1, "reflect"), (2, 256, 64, 12, 40, 1, 1, "zero")]
    for N, Ci, Co, H, W, K, stride, mode in cases:
        x = torch.randn(N, Ci, H, W, device="cuda", requires_grad=True)
        w = (torch.randn(Co, Ci, K, K, device="cuda") * 0.05).requires_grad_(True)
        y = FD.conv2d(x, w, None, stride, K // 2, mode, "none")
        cot = torch.randn_like(y)
        gx, gw = torch.autograd.grad((y * cot).sum(), [x, w])
        xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
        xp = F.pad(xd, (K // 2,) * 4, mode="reflect" if mode == "reflect" else "constant")
        yd = F.conv2d(xp, wd, None, stride)
        gxd, gwd = torch.autograd.grad((yd * cot.double()).sum(), [xd, wd])
        what = "%dx%d k%d s%d %s" % (Ci, Co, K, stride, mode)
        relclose(cpu(y), cpu(yd.float()), "y " + what, rtol=1e-5, arel=3e-6)
        relclose(cpu(gx), cpu(gxd.float()), "dx " + what, rtol=1e-4, arel=1e-5)
        relclose(cpu(gw), cpu(gwd.float()), "dw " + what, rtol=1e-4, arel=1e-5)


@pytest.mark.parametrize("N,C,H,W", [(2, 3, 192, 640), (2, 6, 192, 640), (3, 2, 96, 320), (2, 4, 70, 150), (1, 5, 8, 8), (1, 1, 33, 9)])
def test_stem_convolution_on_the_patch_kernel(FD, N, C, H, W, fdtune):
    """conv_stem.hip (7x7 stride-2 pad-3 stems, networks/resnet_encoder.py:95) against torch float64 conv2d, and against the generic
    gather-GEMM route it replaces (fd_tuning.stem7 = 0): full-size planes, ragged tiles in both directions, odd channel counts
    (padded to a channel pair inside the kernel), planes smaller than one tile."""
    rng = np.random.RandomState(N * 100 + C)
    x = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32))
    w = torch.from_numpy((rng.randn(64, C, 7, 7) * np.sqrt(2.0 / (49 * C))).astype(np.float32))
    ref = F.conv2d(x.double(), w.double(), None, 2, 3).float()
    out = {}
    for on in (1, 0):
        fdtune.lib(stem7=on)
        with torch.no_grad():
            out[on] = FD.conv2d(dev(x), dev(w), None, 2, 3)
        relclose(cpu(out[on]), cpu(ref), "stem forward (stem7=%d)" % on, rtol=1e-5, arel=3e-6)
    relclose(cpu(out[1]), cpu(out[0]), "patch kernel vs gather GEMM", rtol=1e-5, arel=2e-6)


@pytest.mark.parametrize("case", [
    (2, 64, 16, 24, 64, 3, 1, 1, "zero"),        # Winograd data gradient, split-K finish
    (12, 64, 48, 160, 64, 3, 1, 1, "zero"),      # Winograd data gradient, direct epilogue (layer1 at the bench batch)
    (2, 64, 16, 24, 128, 3, 2, 1, "zero"),       # stride 2: four parity classes, each adds its own elements
    (2, 64, 15, 23, 128, 3, 2, 1, "zero"),       # odd sizes
    (2, 64, 16, 24, 128, 1, 2, 0, "zero"),       # 1x1 stride 2: classes without taps -> element-wise fallback
    (2, 64, 8, 12, 256, 1, 1, 0, "zero"),        # bottleneck conv1 (1x1): MFMA GEMM epilogue
    (2, 48, 16, 24, 32, 3, 1, 1, "zero"),        # < 64 output channels: direct kernel
    (2, 96, 16, 24, 32, 3, 1, 1, "reflect"),     # reflect padding: fold pass, then element-wise fallback
    (2, 5, 16, 24, 7, 3, 1, 1, "zero"),          # generic gather GEMM: fallback
])
def test_conv_tap_adds_the_second_gradient_in_the_epilogue(FD, case):
    """FD.conv2d_tap (fd_conv2d_bwd_data_add): d/dx of  sum(conv(x) * a) + sum(x_tap * b)  must be the BITWISE sum torch forms from
    the plain data gradient and b, on every kernel family the data gradient can take."""
    N, Cin, H, W, Cout, K, stride, pad, mode = case
    import zlib
    rng = np.random.RandomState(zlib.crc32(repr(case).encode()) % (2 ** 31))
    x = dev(torch.from_numpy(rng.randn(N, Cin, H, W).astype(np.float32)))
    w = dev(torch.from_numpy((rng.randn(Cout, Cin, K, K) * np.sqrt(2.0 / (Cin * K * K))).astype(np.float32)))
    x1, w1 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y1 = FD.conv2d(x1, w1, None, stride, pad, mode)
    a = dev(torch.from_numpy(rng.randn(*y1.shape).astype(np.float32)))
    b = dev(torch.from_numpy(rng.randn(N, Cin, H, W).astype(np.float32)))
    gx_plain, gw_plain = torch.autograd.grad((y1 * a).sum(), [x1, w1])
    x2, w2 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y2, tap = FD.conv2d_tap(x2, w2, None, stride, pad, mode)
    assert torch.equal(y2, y1) and torch.equal(tap, x2) and tap.data_ptr() == x2.data_ptr()
    gx, gw = torch.autograd.grad((y2 * a).sum() + (tap * b).sum(), [x2, w2])
    assert torch.equal(gx, gx_plain + b), "max |diff| %g" % float((gx - (gx_plain + b)).abs().max())
    assert torch.equal(gw, gw_plain)
    # the tap alone, and the conv alone, still give the right thing
    (g_only_tap,) = torch.autograd.grad((FD.conv2d_tap(x2, w2, None, stride, pad, mode)[1] * b).sum(), [x2])
    assert torch.equal(g_only_tap, b)
    (g_only_conv,) = torch.autograd.grad((FD.conv2d_tap(x2, w2, None, stride, pad, mode)[0] * a).sum(), [x2])
    assert torch.equal(g_only_conv, gx_plain)


def test_conv_transpose_detecting(FD):
    """A = I-style check with asymmetric data: a conv whose weight is a one-hot tap must shift/copy channels
    exactly (catches row/col swaps in the MFMA fragment maps bit-exactly)."""
    N, C, H, W = 1, 40, 9, 70
    x = torch.arange(N * C * H * W, dtype=torch.float32).reshape(N, C, H, W).cuda() % 1013
    w = torch.zeros(C, C, 3, 3)
    for co in range(C):
        w[co, (co * 7 + 3) % C, co % 3, (co // 3) % 3] = 1.0
    y = FD.conv2d(x, w.cuda(), None, 1, 1, "zero", "none", False)
    want = F.conv2d(x.cpu(), w, None, 1, 1)
    assert torch.equal(y.cpu(), want)


@pytest.mark.parametrize("N,C,H,W,res,relu", [(2, 64, 16, 24, False, True), (2, 64, 16, 24, True, True),
                                              (6, 128, 3, 5, True, True), (2, 16, 33, 50, False, False),
                                              (1, 64, 96, 320, False, True), (3, 512, 2, 3, True, True)])
def test_batchnorm_train(FD, N, C, H, W, res, relu):
    rng = np.random.RandomState(N * 1000 + C)
    x = torch.from_numpy((rng.randn(N, C, H, W) * 2 + 3 * rng.randn(1, C, 1, 1)).astype(np.float32))
    r = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32)) if res else None
    bn_o = torch.nn.BatchNorm2d(C)
    gin.fill_params(bn_o, 5)
    bn_g = torch.nn.BatchNorm2d(C)
    bn_g.load_state_dict(bn_o.state_dict())
    bn_g.cuda()
    xo = x.clone().requires_grad_(True)
    ro = r.clone().requires_grad_(True) if res else None
    yo = bn_o(xo)
    if res:
        yo = yo + ro
    if relu:
        yo = F.relu(yo)
    cot = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32))
    leaves_o = [xo, bn_o.weight, bn_o.bias] + ([ro] if res else [])
    want = torch.autograd.grad((yo * cot).sum(), leaves_o)
    xg = dev(x).requires_grad_(True)
    rg = dev(r).requires_grad_(True) if res else None
    yg = FD.batch_norm(xg, bn_g, residual=rg, relu=relu)
    got = torch.autograd.grad((yg * dev(cot)).sum(), [xg, bn_g.weight, bn_g.bias] + ([rg] if res else []))
    relclose(cpu(yg), cpu(yo), "bn fwd", arel=2e-5)
    for a, b, nm in zip(got, want, ["gx", "gweight", "gbias", "gres"]):
        relclose(cpu(a), cpu(b), "bn " + nm, rtol=2e-4, arel=2e-4)
    relclose(cpu(bn_g.running_mean), cpu(bn_o.running_mean), "running_mean")
    relclose(cpu(bn_g.running_var), cpu(bn_o.running_var), "running_var")
    assert int(bn_g.num_batches_tracked) == int(bn_o.num_batches_tracked) == 1
    bn_o.eval(), bn_g.eval()
    with torch.no_grad():
        ye_o = F.relu(bn_o(x)) if relu else bn_o(x)
        ye_g = FD.batch_norm(dev(x), bn_g, relu=relu)
    relclose(cpu(ye_g), cpu(ye_o), "bn eval fwd", arel=2e-5)


@pytest.mark.parametrize("N,C,H,W", [(2, 8, 16, 24), (1, 3, 15, 23), (1, 64, 96, 320), (2, 2, 1, 1), (1, 1, 2, 5)])
def test_maxpool(FD, N, C, H, W):
    rng = np.random.RandomState(H * W)
    x = torch.from_numpy(np.maximum(rng.randn(N, C, H, W), 0).astype(np.float32))   # ReLU-like: many exact ties at 0
    xo = x.clone().requires_grad_(True)
    yo = F.max_pool2d(xo, 3, 2, 1)
    cot = torch.from_numpy(rng.randn(*yo.shape).astype(np.float32))
    want = torch.autograd.grad((yo * cot).sum(), xo)[0]
    xg = dev(x).requires_grad_(True)
    yg = FD.max_pool3x3s2(xg)
    got = torch.autograd.grad((yg * dev(cot)).sum(), xg)[0]
    assert torch.equal(yg.cpu(), yo.detach()), "max-pool forward must be exact"
    relclose(cpu(got), cpu(want), "max-pool backward (tie routing = first max)", rtol=1e-6, arel=1e-6)


def test_upcat_upsample_add_mean(FD):
    rng = np.random.RandomState(4)
    N, Ca, Cs, C3, h, w = 2, 5, 7, 6, 6, 9
    mk = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32))
    a, s1, s2, s3 = mk(N, Ca, h, w), mk(N, Cs, 2 * h, 2 * w), mk(N, Cs, 2 * h, 2 * w), mk(N, C3, 2 * h, 2 * w)
    for use in [(True, True, True), (True, False, False), (False, False, False), (True, True, False), (True, False, True)]:
        ts_o = [t.clone().requires_grad_(True) for t in (a, s1, s2, s3)]
        parts = [OL.upsample(ts_o[0])]
        if use[0]:
            parts.append(ts_o[1] + ts_o[2] if use[1] else ts_o[1])
        if use[2]:
            parts.append(ts_o[3])
        yo = torch.cat(parts, 1)
        cot = mk(*yo.shape)
        leaves = [ts_o[0]] + ([ts_o[1]] if use[0] else []) + ([ts_o[2]] if use[1] else []) + ([ts_o[3]] if use[2] else [])
        want = torch.autograd.grad((yo * cot).sum(), leaves)
        ts_g = [dev(t).requires_grad_(True) for t in (a, s1, s2, s3)]
        yg = FD.upsample_concat(ts_g[0], ts_g[1] if use[0] else None, ts_g[2] if use[1] else None, ts_g[3] if use[2] else None)
        lg = [ts_g[0]] + ([ts_g[1]] if use[0] else []) + ([ts_g[2]] if use[1] else []) + ([ts_g[3]] if use[2] else [])
        got = torch.autograd.grad((yg * dev(cot)).sum(), lg)
        assert torch.equal(yg.cpu(), yo.detach()), "upcat fwd %s" % (use,)
        for g1, g2 in zip(got, want):
            relclose(cpu(g1), cpu(g2), "upcat bwd %s" % (use,), rtol=1e-6, arel=1e-6)
    xg = dev(a).requires_grad_(True)
    assert torch.equal(FD.upsample_nearest2x(xg).cpu(), OL.upsample(a))
    assert torch.equal(FD.add(dev(s1), dev(s2)).cpu(), s1 + s2)
    x = mk(3, 12, 6, 20)
    xo = x.clone().requires_grad_(True)
    mo = 0.01 * xo.mean(3).mean(2)
    cot = mk(3, 12)
    want = torch.autograd.grad((mo * cot).sum(), xo)[0]
    xg = dev(x).requires_grad_(True)
    mg = FD.spatial_mean(xg, 0.01)
    got = torch.autograd.grad((mg * dev(cot)).sum(), xg)[0]
    relclose(cpu(mg), cpu(mo), "spatial mean", rtol=1e-5, arel=1e-6)
    relclose(cpu(got), cpu(want), "spatial mean bwd", rtol=1e-6, arel=1e-6)


def test_depth_errors_and_adam(FD, golden):
    g = golden("layers_b2_32x64")
    errs = FD.depth_errors(torch.from_numpy(g["errs_gt"]).cuda(), torch.from_numpy(g["errs_pred"]).cuda())
    assert_close([float(e) for e in errs], g["errs"], rtol=1e-5, atol=0, what="compute_depth_errors vs reference")
    rng = np.random.RandomState(8)
    p0 = torch.from_numpy(rng.randn(10007).astype(np.float32))
    po = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([po], 1.5e-4)
    pg, m, v = dev(p0), torch.zeros(10007).cuda(), torch.zeros(10007).cuda()
    for step in range(1, 6):
        grad = torch.from_numpy(rng.randn(10007).astype(np.float32))
        po.grad = grad.clone()
        opt.step()
        FD.adam_step(pg, dev(grad), m, v, step, 1.5e-4)
    assert_close(cpu(pg), cpu(po), rtol=1e-6, atol=1e-7, what="Adam after 5 steps")


# ------------------------------------------------------------------------------------------------ networks
def _same_weights(mod_g, mod_o, seed):
    gin.fill_params(mod_o, seed)
    mod_g.load_state_dict(mod_o.state_dict())
    return mod_g.cuda()


@pytest.mark.parametrize("layers,kw,cin,B,H,W", [(18, {}, 3, 2, 64, 96), (18, dict(beam_encoder=True), 2, 2, 64, 96),
                                                 (18, dict(num_input_images=2), 6, 2, 64, 96),
                                                 (18, dict(num_input_images=2, beam_encoder=True), 4, 3, 32, 64),
                                                 (18, {}, 3, 2, 128, 192), (50, {}, 3, 2, 128, 192)])
def test_resnet_encoder_fwd_bwd_vs_oracle(NW, layers, kw, cin, B, H, W):
    """Ground truth is the oracle run in float64.  Train-mode BatchNorm over a handful of samples (layer4 sees B*2*3 values per
    channel here) amplifies fp32 rounding, so ANY float32 evaluation of the deepest gradients is only accurate to ~1e-3: the
    yardstick per tensor is the error of two independent float32 evaluations of the same oracle graph - ATen on the CPU (whose
    BatchNorm accumulates in double) and ATen / MIOpen on the GPU.  Every feature map and every parameter gradient of the HIP
    encoder must be within 1e-4 (relative L1) of float64, or within 3x the worse of those two float32 errors FOR THAT TENSOR
    (the two yardsticks themselves differ by up to 2.6x on single tensors of the ResNet-50 case; all errors are printed)."""
    enc_o = ON.ResnetEncoder(layers, False, **kw)
    enc_g = _same_weights(NW.ResnetEncoder(layers, False, **kw), enc_o, 21)
    import copy
    enc_d = copy.deepcopy(enc_o).double()
    enc_a = copy.deepcopy(enc_o).cuda()                      # the oracle's torch graph on the GPU: a second fp32 yardstick
    enc_o.train(), enc_g.train(), enc_d.train(), enc_a.train()
    # seed: a ReLU pre-activation within rounding of zero flips its mask in one float32 implementation and not in another (an
    # O(1e-3) change of every upstream gradient of these tiny tensors; scripts/bn_small_dbg.py shows one such element for seed 17; seeds 17 and 23 each tie somewhere in one of the six cases
    # between two summation orders of the same BatchNorm).  The seed can be overridden to probe for that.
    rng = np.random.RandomState(int(os.environ.get("FD_TEST_SEED", "5")))
    x = torch.from_numpy(rng.rand(B, cin, H, W).astype(np.float32))
    fo, fd_, fg, fa = enc_o(x), enc_d(x.double()), enc_g(dev(x)), enc_a(dev(x))
    cots = [torch.from_numpy(rng.randn(*f.shape).astype(np.float32)) for f in fo]

    def agg(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-30)

    report, bad = [], []
    for i in range(5):
        report.append(("feature%d" % i, agg(cpu(fg[i]), cpu(fd_[i])), agg(cpu(fo[i]), cpu(fd_[i])), agg(cpu(fa[i]), cpu(fd_[i]))))
    keep = lambda m: [(n, p) for n, p in m.named_parameters() if ".fc." not in n]
    names = [n for n, _ in keep(enc_o)]
    want32 = torch.autograd.grad(sum((f * c).sum() for f, c in zip(fo, cots)), [p for _, p in keep(enc_o)])
    want64 = torch.autograd.grad(sum((f * c.double()).sum() for f, c in zip(fd_, cots)), [p for _, p in keep(enc_d)])
    wantgpu = torch.autograd.grad(sum((f * dev(c)).sum() for f, c in zip(fa, cots)), [p for _, p in keep(enc_a)])
    got = torch.autograd.grad(sum((f * dev(c)).sum() for f, c in zip(fg, cots)), [p for _, p in keep(enc_g)])
    for n, a, b32, bgpu, b64 in zip(names, got, want32, wantgpu, want64):
        report.append((n, agg(cpu(a), cpu(b64)), agg(cpu(b32), cpu(b64)), agg(cpu(bgpu), cpu(b64))))
    for n, e_hip, e_cpu, e_gpu in report:
        if e_hip > max(1e-4, 3 * max(e_cpu, e_gpu)):
            bad.append("%s: HIP err %.3g vs float32 oracle err %.3g (CPU) / %.3g (GPU)" % (n, e_hip, e_cpu, e_gpu))
    worst = max(report, key=lambda r: r[1])
    print("worst: %s HIP %.3g (float32 oracle: CPU %.3g, GPU %.3g); %d of %d tensors above 1e-4" % (
        worst + (sum(r[1] > 1e-4 for r in report), len(report))))
    assert not bad, "\n".join(bad[:20])
    sd_o, sd_g = enc_d.state_dict(), enc_g.state_dict()
    for k in sd_o:
        if "running" in k:
            relclose(cpu(sd_g[k]), cpu(sd_o[k]), k, rtol=2e-4, arel=2e-4)


@pytest.mark.parametrize("kind,cin,planes,stride", [("basic", 64, 64, 1), ("basic", 64, 128, 2), ("bottleneck", 64, 64, 1),
                                                    ("bottleneck", 256, 128, 2), ("bottleneck", 512, 128, 1)])
def test_residual_blocks_vs_fp64(kind, cin, planes, stride):
    """Single ResNet blocks (well conditioned: 2*12*18 samples per BN channel) must be accurate to ~1e-5 against
    a float64 oracle — separates kernel correctness from the deep-net rounding amplification seen above."""
    from fusiondepth_amd.networks import resnet_encoder as RE
    from oracle import networks as ONN
    blk_o = (ONN._Basic if kind == "basic" else ONN._Bottle)(cin, planes, stride)
    blk_g = (RE.BasicBlock if kind == "basic" else RE.Bottleneck)(cin, planes, stride)
    gin.fill_params(blk_o, 33)
    blk_g.load_state_dict(blk_o.state_dict())
    blk_g.cuda().train()
    blk_d = __import__("copy").deepcopy(blk_o).double().train()
    rng = np.random.RandomState(2)
    x = torch.from_numpy(np.maximum(rng.randn(2, cin, 12, 18), 0).astype(np.float32))
    xd, xg = x.double().requires_grad_(True), dev(x).requires_grad_(True)
    yd, yg = blk_d(xd), blk_g(xg)
    cot = torch.from_numpy(rng.randn(*yd.shape).astype(np.float32))
    want = torch.autograd.grad((yd * cot.double()).sum(), [xd] + list(blk_d.parameters()))
    got = torch.autograd.grad((yg * dev(cot)).sum(), [xg] + list(blk_g.parameters()))
    relclose(cpu(yg), cpu(yd), "block fwd", rtol=1e-5, arel=1e-5)
    for n, a, b in zip(["x"] + [k for k, _ in blk_g.named_parameters()], got, want):
        a, b = cpu(a).astype(np.float64), cpu(b)
        err = np.abs(a - b).sum() / max(np.abs(b).sum(), 1e-30)
        assert err < 2e-5, "grad %s: aggregate relative error %.3g" % (n, err)


def test_depth_and_pose_decoders_vs_reference_golden(NW, golden):
    g = golden("decoders_b2_64x96")
    B, H, W = 2, 64, 96
    rng = np.random.RandomState(202)
    ch = np.array([64, 64, 128, 256, 512])
    feats, beams = gin.feature_pyramids(rng, B, H, W, ch)
    fg = [dev(t).requires_grad_(True) for t in feats]
    bg = [dev(t).requires_grad_(True) for t in beams]
    dec = gin.fill_params(NW.DepthDecoder(ch), 11).cuda()
    o = dec(fg, beam_features=bg)
    for s in range(4):
        relclose(cpu(o[("disp", s)]), g["dec_disp%d" % s], "disp%d vs reference" % s)
    loss = sum((o[("disp", s)] * torch.from_numpy(g["dec_cot%d" % s]).cuda()).sum() for s in range(4))
    grads = torch.autograd.grad(loss, fg + bg + list(dec.parameters()))
    for i in range(5):
        check_grad_compact(g, "dec_gfeat%d" % i, cpu(grads[i]), rtol=1e-3, atol=1e-4 * float(np.abs(cpu(grads[i])).max()))
        check_grad_compact(g, "dec_gbeam%d" % i, cpu(grads[5 + i]), rtol=1e-3, atol=1e-4 * float(np.abs(cpu(grads[5 + i])).max()))
    for (k, _), gv in zip(dec.named_parameters(), grads[10:]):
        check_grad_compact(g, "dec_g/" + k.replace(".", "/"), cpu(gv), rtol=1e-3, atol=1e-4 * float(np.abs(cpu(gv)).max()))
    o2 = dec([dev(t) for t in feats])
    relclose(cpu(o2[("disp", 0)]), g["dec_nobeam_disp0"], "no-beam disp0 vs reference")

    pose = gin.fill_params(NW.PoseDecoder(ch, num_input_features=1, num_frames_to_predict_for=2), 12).cuda()
    f4, b4 = dev(feats[4]).requires_grad_(True), dev(beams[4]).requires_grad_(True)
    aa, tr = pose([[None] * 4 + [f4]], beam_inputs=[[None] * 4 + [b4]])
    relclose(cpu(aa), g["pose_aa"], "axisangle vs reference")
    relclose(cpu(tr), g["pose_tr"], "translation vs reference")
    gr = torch.autograd.grad((aa * torch.from_numpy(g["pose_cot_a"]).cuda()).sum() +
                             (tr * torch.from_numpy(g["pose_cot_t"]).cuda()).sum(), [f4, b4] + list(pose.parameters()))
    relclose(cpu(gr[0]), g["pose_gf4"], "pose g f4 vs reference", rtol=1e-3)
    relclose(cpu(gr[1]), g["pose_gb4"], "pose g b4 vs reference", rtol=1e-3)
    for (k, _), gv in zip(pose.named_parameters(), gr[2:]):
        check_grad_compact(g, "pose_g/" + k.replace(".", "/"), cpu(gv), rtol=1e-3, atol=1e-4 * float(np.abs(cpu(gv)).max()))

    pcnn = gin.fill_params(NW.PoseCNN(2), 13).cuda()
    xin = torch.from_numpy(np.random.RandomState(213).rand(B, 6, H, W).astype(np.float32)).cuda()
    a2, t2 = pcnn(xin)
    relclose(cpu(a2), g["posecnn_aa"], "PoseCNN axisangle vs reference")
    relclose(cpu(t2), g["posecnn_tr"], "PoseCNN translation vs reference")


def test_refine_decoder_variant_vs_reference_golden(NW, golden):
    g = golden("decoder_refine_b2_64x96")
    B, H, W = 2, 64, 96
    rng = np.random.RandomState(202)
    ch = np.array([64, 64, 128, 256, 512])
    feats, beams = gin.feature_pyramids(rng, B, H, W, ch)
    dec2 = gin.fill_params(NW.DepthDecoder(ch, road=True, catxy=True, deep=True), 14).cuda()
    rng2 = np.random.RandomState(214)
    dm = {("disp", s): torch.from_numpy(rng2.rand(B, 6, H // 2 ** s, W // 2 ** s).astype(np.float32)).cuda() for s in range(4)}
    o = dec2([dev(t) for t in feats], beam_features=[dev(t) for t in beams], depth_maps=dm, tanh=True)
    for s in range(4):
        relclose(cpu(o[("disp", s)]), g["disp%d" % s], "refine disp%d vs reference" % s)


def test_layers_module_api(golden):
    """Drop-in ``layers`` classes keep the reference's constructor/forward signatures."""
    from fusiondepth_amd import layers as L
    g = golden("layers_b2_32x64")
    B, H, W = 2, 32, 64
    inp, rng = gin.batch_inputs(101, B, H, W)
    depth = torch.from_numpy(g["d2d_depth"]).cuda()
    pts = L.BackprojectDepth(B, H, W)(depth, dev(inp[("inv_K", 0)]))
    grid = L.Project3D(B, H, W)(pts, dev(inp[("K", 0)]), torch.from_numpy(g["pj_T"]).cuda())
    relclose(cpu(grid), g["pj_grid"], "layers.Project3D")
    cb = L.ConvBlock(5, 7).cuda()
    with torch.no_grad():
        cb.conv.conv.weight.copy_(torch.from_numpy(g["cb_w"]))
        cb.conv.conv.bias.copy_(torch.from_numpy(g["cb_b"]))
    relclose(cpu(cb(torch.from_numpy(g["cb_x"]).cuda())), g["cb_y"], "layers.ConvBlock vs reference")
    c3 = L.Conv3x3(5, 1, use_refl=False).cuda()
    with torch.no_grad():
        c3.conv.weight.copy_(torch.from_numpy(g["c3_w"]))
        c3.conv.bias.copy_(torch.from_numpy(g["c3_b"]))
    relclose(cpu(c3(torch.from_numpy(g["cb_x"]).cuda())), g["c3_y"], "layers.Conv3x3(zero pad) vs reference")
    relclose(cpu(L.upsample(torch.from_numpy(g["cb_x"]).cuda())), g["up_y"], "layers.upsample")
    relclose(cpu(L.SSIM()(dev(inp[("color", 0, 0)]), dev(inp[("color", 1, 0)]))), g["ssim"], "layers.SSIM", arel=2e-5)
    relclose(cpu(L.Cat_xy(B, H, W)(depth, dev(inp[("inv_K", 0)]))), g["catxy"], "layers.Cat_xy")
    assert list(cb.state_dict().keys()) == ["conv.conv.weight", "conv.conv.bias"]


@pytest.mark.parametrize("N,C,H,W,G,res", [(4, 64, 12, 18, 2, True), (12, 128, 3, 5, 2, False), (6, 16, 9, 7, 3, True)])
def test_grouped_batchnorm_equals_separate_passes(FD, N, C, H, W, G, res):
    """FD.bn_groups(G): one launch over a stacked batch == G consecutive forward passes (statistics, outputs, gradients,
    and the in-order momentum updates of the running statistics)."""
    rng = np.random.RandomState(N + C)
    x = torch.from_numpy((rng.randn(N, C, H, W) * 2 + rng.randn(N, 1, 1, 1)).astype(np.float32))
    r = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32)) if res else None
    cot = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32))
    bn_o = gin.fill_params(torch.nn.BatchNorm2d(C), 5)
    bn_g = torch.nn.BatchNorm2d(C)
    bn_g.load_state_dict(bn_o.state_dict())
    bn_g.cuda()
    xo = x.clone().requires_grad_(True)
    ro = r.clone().requires_grad_(True) if res else None
    Ng = N // G
    outs = []
    for g in range(G):
        y = bn_o(xo[g * Ng:(g + 1) * Ng])
        if res:
            y = y + ro[g * Ng:(g + 1) * Ng]
        outs.append(F.relu(y))
    yo = torch.cat(outs, 0)
    want = torch.autograd.grad((yo * cot).sum(), [xo, bn_o.weight, bn_o.bias] + ([ro] if res else []))
    xg = dev(x).requires_grad_(True)
    rg = dev(r).requires_grad_(True) if res else None
    with FD.bn_groups(G):
        yg = FD.batch_norm(xg, bn_g, residual=rg, relu=True)
    got = torch.autograd.grad((yg * dev(cot)).sum(), [xg, bn_g.weight, bn_g.bias] + ([rg] if res else []))
    relclose(cpu(yg), cpu(yo), "grouped bn fwd", arel=2e-5)
    for a, b, nm in zip(got, want, ["gx", "gweight", "gbias", "gres"]):
        relclose(cpu(a), cpu(b), "grouped bn " + nm, rtol=2e-4, arel=2e-4)
    relclose(cpu(bn_g.running_mean), cpu(bn_o.running_mean), "running_mean after %d in-order updates" % G)
    relclose(cpu(bn_g.running_var), cpu(bn_o.running_var), "running_var after %d in-order updates" % G)
    assert int(bn_g.num_batches_tracked) == int(bn_o.num_batches_tracked) == G


@pytest.mark.parametrize("kern", ["1d", "2p"])
@pytest.mark.parametrize("N,Cin,Cout,H,W,G,res", [
    (6, 64, 64, 48, 160, 2, True),      # the step's layer1 plane: 60 tiles per image, grouped, residual + ReLU
    (12, 128, 128, 24, 80, 3, False),   # layer2: 15 tiles per image, 2 channel tiles, three groups
    (8, 128, 64, 48, 160, 1, True),     # one group
])
def test_conv_epilogue_statistics_feed_batchnorm(FD, N, Cin, Cout, H, W, G, res, kern, fdtune):
    """fd_conv2d_fwd_stats + fd_bn_train_fwd_parts (BatchNorm statistics gathered in the convolution's epilogue, one BatchNorm
    launch) == fd_conv2d_fwd + fd_bn_train_fwd (statistics pass + apply pass), and both == torch's conv2d + BatchNorm2d in float64:
    outputs, running statistics after the grouped in-order updates, and every gradient.  Reference ops: torchvision BasicBlock
    (conv3x3 -> BatchNorm2d -> [+ identity] -> ReLU) as driven by networks/resnet_encoder.py:95-101."""
    # kern: "1d" = k_conv_wino (slots of 64 pixels), "2p" = k_conv_wino2p (slots of 32 2x2 tiles = 128 pixels; a plane whose tile count
    # is 64 k + 32 - the 24x80 case, ResNet layer2 at 640x192 - is tiled image by image, its last tile half empty)
    fdtune.lib(wino_fwd_2dp_min_wgs=1 if kern == "2p" else 0)
    slots = 2 * (H * W // 2) // 64 if kern == "1d" else ((H * W // 4) // 32 if (H * W // 4) % 32 == 0 else None)
    rng = np.random.RandomState(N * 1000 + Cin)
    x = torch.from_numpy((rng.randn(N, Cin, H, W) + 0.5).astype(np.float32))
    w = torch.from_numpy((rng.randn(Cout, Cin, 3, 3) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32))
    r = torch.from_numpy(rng.randn(N, Cout, H, W).astype(np.float32)) if res else None
    cot = torch.from_numpy(rng.randn(N, Cout, H, W).astype(np.float32))
    bn_ref = gin.fill_params(torch.nn.BatchNorm2d(Cout), 7)
    Ng = N // G

    def hip(use_stats):
        bn = torch.nn.BatchNorm2d(Cout)
        bn.load_state_dict(bn_ref.state_dict())
        bn.cuda()
        xg, wg = dev(x).requires_grad_(True), dev(w).requires_grad_(True)
        rg = dev(r).requires_grad_(True) if res else None
        with FD.bn_groups(G):
            if use_stats:
                y, stats = FD.conv2d_stats(xg, wg, None, 1, 1)
                assert (stats is None) if slots is None else (stats is not None and tuple(stats.shape) == (N, Cout, slots, 2))
                out = FD.batch_norm(y, bn, residual=rg, relu=True, conv_stats=stats)
            else:
                out = FD.batch_norm(FD.conv2d(xg, wg, None, 1, 1), bn, residual=rg, relu=True)
        gr = torch.autograd.grad((out * dev(cot)).sum(), [xg, wg, bn.weight, bn.bias] + ([rg] if res else []))
        return out, gr, bn

    out_s, gr_s, bn_s = hip(True)
    out_p, gr_p, bn_p = hip(False)
    # float64 ground truth, group by group
    bn64 = torch.nn.BatchNorm2d(Cout).double()
    bn64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn_ref.state_dict().items()})
    x64, w64 = x.double().requires_grad_(True), w.double().requires_grad_(True)
    r64 = r.double().requires_grad_(True) if res else None
    outs = []
    for g in range(G):
        y = bn64(F.conv2d(x64[g * Ng:(g + 1) * Ng], w64, None, 1, 1))
        if res:
            y = y + r64[g * Ng:(g + 1) * Ng]
        outs.append(F.relu(y))
    out64 = torch.cat(outs, 0)
    gr64 = torch.autograd.grad((out64 * cot.double()).sum(), [x64, w64, bn64.weight, bn64.bias] + ([r64] if res else []))
    from conftest import assert_mostly_close
    names = ["gx", "gw", "g gamma", "g beta", "g residual"]
    for name, out, gr, bn in (("epilogue statistics", out_s, gr_s, bn_s), ("statistics pass", out_p, gr_p, bn_p)):
        relclose(cpu(out), out64.detach().numpy(), name + ": output", arel=2e-5)
        relclose(cpu(bn.running_mean), bn64.running_mean.numpy(), name + ": running_mean")
        relclose(cpu(bn.running_var), bn64.running_var.numpy(), name + ": running_var")
        for a, b, nm in zip(gr, gr64, names):
            b = b.numpy()
            # An output within float32 rounding of 0 takes the other side of the ReLU than in float64: its gradient (and through
            # the 3x3 window that of its neighbours) flips by O(1) - a few 1e-4 of the entries at these sizes; the parameter
            # gradients are sums over all pixels and move by those flipped entries (a percent of their largest value).
            if nm in ("gx", "g residual"):
                assert_mostly_close(cpu(a), b, rtol=2e-4, atol=2e-4 * np.abs(b).max(), what="%s: %s" % (name, nm), max_bad_frac=2e-3)
            else:
                relclose(cpu(a), b, "%s: %s" % (name, nm), rtol=1e-3, arel=2e-2)
    # the two HIP paths differ only in the summation order of the statistics (1e-7 relative in mean / variance)
    relclose(cpu(out_s), cpu(out_p), "epilogue statistics vs statistics pass: output", arel=2e-6)
    for a, b, nm in zip(gr_s, gr_p, names):
        if nm in ("gx", "g residual"):
            assert_mostly_close(cpu(a), cpu(b), rtol=2e-4, atol=2e-4 * float(b.abs().max()), what="two paths: " + nm, max_bad_frac=1e-3)
        else:
            relclose(cpu(a), cpu(b), "two paths: " + nm, rtol=1e-3, arel=5e-3)


def test_batchnorm_statistics_of_a_near_constant_map(FD):
    """The beam encoders see a sparse LiDAR image: after input normalisation and a zero-padded convolution the map is one constant
    (a different one along the border) plus sparse spikes, i.e. the channel mean sits 10 - 100 standard deviations from zero and
    from the corner pixel.  Single-pass E[d^2] - E[d]^2 statistics shifted by the corner pixel (rounds 1-2) lost 3 digits of the
    variance there (2.4e-4 on the beam encoder's first feature at 1216x352).  Both statistics paths - fd_bn_train_fwd's own pass and
    the convolution-epilogue partials of fd_conv2d_fwd_stats - against float64 BatchNorm2d on such a map, bound 2e-5."""
    rng = np.random.RandomState(11)
    N, C, H, W = 2, 64, 96, 320
    x = np.full((N, C, H, W), -2.0, np.float32)                          # (0 - 0.45) / 0.225
    hit = rng.rand(N, 1, H, W) < 0.03
    x = np.where(hit, (rng.rand(N, C, H, W) * 4 - 2).astype(np.float32), x).astype(np.float32)
    x = torch.from_numpy(x)
    w = torch.from_numpy((np.abs(rng.randn(C, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(np.float32))    # same-sign taps: |mean| >> std
    bn_ref = gin.fill_params(torch.nn.BatchNorm2d(C), 9)
    bn64 = torch.nn.BatchNorm2d(C).double()
    bn64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in bn_ref.state_dict().items()})
    y64 = F.conv2d(x.double(), w.double(), None, 1, 1)
    ratio = float((y64.mean((0, 2, 3)).abs() / y64.std((0, 2, 3))).median())
    assert ratio > 8, ratio                                              # the premise of the test
    out64 = bn64(y64)
    for use_stats in (True, False):
        bn = torch.nn.BatchNorm2d(C)
        bn.load_state_dict(bn_ref.state_dict())
        bn.cuda()
        with torch.no_grad():
            if use_stats:
                y, stats = FD.conv2d_stats(dev(x), dev(w), None, 1, 1)
                assert stats is not None
                out = FD.batch_norm(y, bn, conv_stats=stats)
            else:
                out = FD.batch_norm(FD.conv2d(dev(x), dev(w), None, 1, 1), bn)
        name = "near-constant map, " + ("epilogue statistics" if use_stats else "statistics pass")
        relclose(cpu(out), out64.detach().numpy(), name + ": output", arel=2e-5)
        relclose(cpu(bn.running_var), bn64.running_var.numpy(), name + ": running_var", rtol=2e-5, arel=2e-5)
        relclose(cpu(bn.running_mean), bn64.running_mean.numpy(), name + ": running_mean", rtol=2e-6, arel=2e-6)


@pytest.mark.parametrize("N,Ci,Co,H,W", [(2, 64, 32, 3, 6), (1, 128, 64, 2, 4), (2, 96, 32, 5, 8), (1, 16, 16, 4, 10), (2, 64, 64, 12, 40),
                                         (1, 32, 48, 3, 3), (1, 512, 256, 6, 20), (2, 32, 16, 7, 5),
                                         (2, 96, 32, 96, 320), (1, 16, 16, 192, 640), (1, 288, 32, 96, 320),      # >= 16 384 pixels: the ring path by default
                                         (2, 512, 256, 12, 40), (2, 128, 64, 48, 160), (3, 64, 32, 5, 6)])       # upconv(4,1), upconv(2,1); odd height
@pytest.mark.parametrize("wino", [1, 0, "padded"])
def test_reflect_padded_data_gradient_by_interior_plus_ring(FD, N, Ci, Co, H, W, wino, fdtune):
    """conv3x3(ReflectionPad2d(1)(x)) - every DepthDecoder convolution (networks/depth_decoder.py, layers.py Conv3x3): its data
    gradient = the zero-padded data gradient written straight to gx + the padded grid's one-pixel ring (four strips, one grouped
    launch) folded back onto rows 1 / H-2 and columns 1 / W-2 (k_reflect_ring_fold), including images so small that the two target
    rows / columns coincide (H = 3) or are the border itself (H = 2), odd sizes, and a second gradient joining at the input
    (conv2d_tap).  wino = 1 (default): where the input has >= 64 channels and the width is even the interior runs on the Winograd
    kernels (F(2, 3) per kernel row, F(2x2, 3x3) from 256 x 256 channels on) with its own weight layout behind the ring's; 0: the
    interior on the implicit-GEMM kernel everywhere; "padded": the default routing - planes below 16 384 pixels with >= 64 input
    channels take the whole padded-grid gradient as one Winograd convolution over dY in a border of zeros + the fold pass.
    Against torch's float64 autograd of F.pad(mode="reflect") + conv2d."""
    fdtune.lib(reflect_wino=0 if wino == 0 else 1)
    ring = lambda: fdtune.lib(reflect_ring=1 if wino == "padded" else 2)
    ring()                                            # 2: planes from 2 pixels on (default 1: from 16 384 - smaller ones keep a fold pass)
    g = torch.Generator().manual_seed(N * 131 + H)
    x = torch.randn(N, Ci, H, W, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) * (2.0 / (9 * Ci)) ** 0.5
    cot = torch.randn(N, Co, H, W, generator=g)
    x64 = x.double().requires_grad_(True)
    y64 = F.conv2d(F.pad(x64, (1, 1, 1, 1), mode="reflect"), w.double())
    gx64 = torch.autograd.grad((y64 * cot.double()).sum(), x64)[0].numpy()
    xg = dev(x).requires_grad_(True)
    y = FD.conv2d(xg, dev(w), None, 1, 1, "reflect")
    relclose(cpu(y), y64.detach().numpy(), "forward", arel=2e-6)
    gx = torch.autograd.grad((y * dev(cot)).sum(), xg)[0]
    relclose(cpu(gx), gx64, "data gradient", arel=3e-6)
    fdtune.lib(reflect_ring=0)                        # the fold path gives the same gradient up to the order of the ring's additions
    gx_fold = torch.autograd.grad((FD.conv2d(xg, dev(w), None, 1, 1, "reflect") * dev(cot)).sum(), xg)[0]
    relclose(cpu(gx), cpu(gx_fold), "ring path vs fold path", arel=1e-6)
    ring()
    xg2 = dev(x).requires_grad_(True)
    y2, xt = FD.conv2d_tap(xg2, dev(w), None, 1, 1, "reflect")
    gx2 = torch.autograd.grad((y2 * dev(cot)).sum() + (xt * xt).sum(), xg2)[0]
    relclose(cpu(gx2), gx64 + 2 * x.numpy(), "data gradient + tap", arel=3e-6)


def test_conv_epilogue_statistics_decline_unsupported_shapes(FD):
    """Shapes whose kernel has no statistics epilogue (tiles straddling images, split-K, strided / 1x1 convolutions on the direct
    kernel): conv2d_stats returns None and batch_norm makes its own statistics pass."""
    for (N, Cin, Cout, H, W, k, s, p) in [(2, 64, 64, 6, 20, 3, 1, 1), (2, 64, 128, 16, 32, 3, 2, 1), (2, 64, 64, 16, 32, 1, 1, 0)]:
        x = torch.randn(N, Cin, H, W, device="cuda")
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
        y, stats = FD.conv2d_stats(x, w, None, s, p)
        assert stats is None
        assert torch.equal(y, FD.conv2d(x, w, None, s, p))


# ------------------------------------------------------------------------------------------------ Winograd F(2,3) path
def _wino_conv(x, w, bias, reflect, act=0):
    import ctypes
    from fusiondepth_amd import _lib
    N, C, H, W = x.shape
    d = _lib.ConvDesc(N, C, H, W, w.shape[0], 3, 3, 1, 1, 1 if reflect else 0, act, 0)
    nwt = _lib.query("fd_conv3x3_wino_wt_floats", ctypes.byref(d))
    assert nwt in (4 * w.shape[0] * 3 * C, 4 * w.shape[0] * 4 * C)          # F(2, 3) per kernel row / F(2x2, 3x3)
    y = torch.empty(N, w.shape[0], H, W, device="cuda")
    wt = torch.empty(nwt, device="cuda")
    ws = torch.empty(max(_lib.query("fd_conv3x3_wino_ws_floats", ctypes.byref(d)), 1), device="cuda")
    _lib.call("fd_conv3x3_wino_fwd", ctypes.byref(d), x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
              y.data_ptr(), wt.data_ptr(), 0, ws.data_ptr(), _lib.stream())
    return y, nwt


@pytest.mark.parametrize("two_d", [1, 0, 2, 3])
@pytest.mark.parametrize("N,Ci,Co,H,W,reflect,act", [
    (2, 64, 64, 48, 160, False, 0),      # layer1 shape, no split
    (1, 256, 256, 12, 40, False, 1),     # few tiles: split-K slabs + finish (bias + ReLU applied there)
    (2, 96, 80, 7, 10, False, 0),        # 80 output channels (partial channel tile), odd height, pairs < one tile
    (2, 96, 80, 8, 10, False, 0),        # ... the same with whole 2x2 tiles
    (1, 128, 64, 24, 80, True, 2),       # reflect padding (decoder ConvBlock), ELU
    (3, 16, 64, 5, 6, True, 0),          # 16 input channels = a single chunk per kernel row
    (3, 16, 64, 6, 6, True, 3),          # ... every 2x2 tile at one or two borders, sigmoid
    (2, 64, 64, 2, 4, True, 0),          # one tile row: both vertical mirrors in the same tile
    (2, 64, 64, 2, 4, False, 0),
    (3, 512, 512, 6, 20, False, 1),      # layer4: the 2-D kernel also splits the input channels
    (2, 96, 32, 24, 40, True, 2),        # 32 output channels (upconv(1,1) of the depth decoder): two_d = 2 is k_conv_wino2p_dma<false, HALFM>
    (3, 64, 32, 12, 20, False, 1),       # ... zero padding, pixel tiles across image borders
    (2, 32, 16, 8, 12, True, 0),         # 16 output channels, two chunks per row component
])
def test_winograd_conv_vs_float64_reference(N, Ci, Co, H, W, reflect, act, two_d, fdtune):
    """conv_wino.hip through its own entry point against torch float64 conv2d: error within a few fp32 ulps of the output scale,
    i.e. no worse than the direct implicit GEMM (transform coefficients are +-1 and 1/2).  two_d = 1: F(2x2, 3x3) (k_conv_wino2d +
    k_wino2d_finish) forced onto every shape with an even height; 2: F(2x2, 3x3) with the 16 components in one workgroup
    (k_conv_wino2p, round 4) forced likewise; 0: F(2, 3) per kernel row everywhere."""
    # (3: k_conv_wino2p with the register-staged loader - what widths that are not multiples of 4 take - instead of the direct-to-LDS one)
    fdtune.lib(wino_fwd_2d_min=1 if two_d == 1 else 0, wino_fwd_2dp_min_wgs=1 if two_d >= 2 else 0, wino_fwd_2dp_dma=0 if two_d == 3 else 1)
    torch.manual_seed(N * 1000 + Ci)
    x = torch.randn(N, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
    b = torch.randn(Co, device="cuda")
    xp = F.pad(x.double(), (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    ref = F.conv2d(xp, w.double(), b.double())
    ref = {0: ref, 1: F.relu(ref), 2: F.elu(ref), 3: torch.sigmoid(ref)}[act]
    got, nwt = _wino_conv(x, w, b, reflect, act)
    assert nwt == 4 * Co * (4 if two_d and H % 2 == 0 else 3) * Ci
    relclose(cpu(got), cpu(ref.float()), "winograd conv", rtol=1e-5, arel=3e-6)


@pytest.mark.parametrize("m128", [1, 0])
@pytest.mark.parametrize("N,Ci,Co,H,W,reflect,act", [
    (1, 256, 256, 12, 40, False, 1),     # layer3: channel splits on top of the four row components
    (3, 512, 512, 6, 20, False, 1),      # layer4
    (2, 256, 128, 24, 80, True, 2),      # upconv(3, 1) of the depth decoder: reflect padding, ELU in the finish pass
    (3, 80, 128, 12, 20, False, 0),      # 10 chunks of 8 channels (uneven splits), 180 tiles = 2.8 pixel tiles across image borders
    (2, 128, 128, 6, 8, True, 3),        # every tile at a border, rows of two 16-byte pieces
    (3, 64, 128, 2, 4, False, 0),        # one tile row per image
    (2, 64, 384, 8, 12, False, 0),       # three channel tiles of 128
    (2, 64, 128, 32, 64, False, 1),      # 16 pixel tiles x 1 channel tile x 4 components: the 3-D grid with the XCD swizzle of the pixel tiles
])
def test_winograd_slab_kernel_with_128_channel_tiles(N, Ci, Co, H, W, reflect, act, m128, fdtune):
    """k_conv_wino2d_m128 (128 output channels per workgroup, activations direct-to-LDS, 8-channel chunks) against torch float64,
    wherever it can run (fd_tuning.wino_fwd_2d_m128 = 1, the default), next to k_conv_wino2d (0) on the same shapes: both must sit within a few fp32 ulps of the output
    scale, and within that of each other."""
    fdtune.lib(wino_fwd_2d_min=1, wino_fwd_2dp_min_wgs=0, wino_fwd_2d_m128=m128)
    torch.manual_seed(N * 1000 + Ci + Co)
    x = torch.randn(N, Ci, H, W, device="cuda")
    w = torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05
    b = torch.randn(Co, device="cuda")
    xp = F.pad(x.double(), (1, 1, 1, 1), mode="reflect" if reflect else "constant")
    ref = F.conv2d(xp, w.double(), b.double())
    ref = {0: ref, 1: F.relu(ref), 2: F.elu(ref), 3: torch.sigmoid(ref)}[act]
    got, nwt = _wino_conv(x, w, b, reflect, act)
    assert nwt == 4 * Co * 4 * Ci
    relclose(cpu(got), cpu(ref.float()), "winograd slabs (m128=%d)" % m128, rtol=1e-5, arel=3e-6)


@pytest.mark.parametrize("mode", ["zero", "reflect"])
def test_one_workgroup_winograd_forward_backward_with_residual_tap(mode, fdtune):
    """k_conv_wino2p through FD.conv2d / FD.conv2d_tap: forward (bias + ReLU in its epilogue) and, for zero padding, the data
    gradient with the second incoming gradient added in the epilogue (fd_conv2d_bwd_data_add: both rows of every 2x2 tile),
    against torch float64 autograd; layer1-like plane with pixel tiles that cross image borders (3 x 12 x 20 = 720 tiles)."""
    import fusiondepth_amd.functional as FD
    fdtune.lib(wino_fwd_2dp_min_wgs=1)
    torch.manual_seed(77)
    x = torch.randn(3, 64, 24, 40, device="cuda", requires_grad=True)
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
    b = torch.randn(64, device="cuda") * 0.1
    y, xt = FD.conv2d_tap(x, w, b, 1, 1, mode, "relu")
    cot = torch.randn_like(y)
    gx = torch.autograd.grad((y * cot).sum() + (xt * xt).sum(), x)[0]
    xd = x.detach().double().requires_grad_(True)
    xp = F.pad(xd, (1, 1, 1, 1), mode="reflect" if mode == "reflect" else "constant")
    yd = F.relu(F.conv2d(xp, w.double(), b.double()))
    gxd = torch.autograd.grad((yd * cot.double()).sum() + (xd * xd).sum(), xd)[0]
    relclose(cpu(y), cpu(yd.float()), "y", rtol=1e-5, arel=3e-6)
    from conftest import assert_mostly_close
    assert_mostly_close(cpu(gx), cpu(gxd.float()), rtol=1e-4, atol=1e-4 * float(gxd.abs().max()), what="dx + tap", max_bad_frac=1e-3)


def test_winograd_refuses_ineligible_shapes():
    import ctypes
    from fusiondepth_amd import _lib
    for desc in (_lib.ConvDesc(1, 64, 8, 9, 64, 3, 3, 1, 1, 0, 0, 0),      # odd width
                 _lib.ConvDesc(1, 24, 8, 8, 64, 3, 3, 1, 1, 0, 0, 0),      # Cin not a multiple of 16
                 _lib.ConvDesc(1, 64, 8, 8, 64, 3, 3, 2, 1, 0, 0, 0)):     # stride 2
        assert _lib.query("fd_conv3x3_wino_wt_floats", ctypes.byref(desc)) == 0
    d = _lib.ConvDesc(1, 64, 8, 9, 64, 3, 3, 1, 1, 0, 0, 0)
    t = torch.zeros(64 * 64 * 9 * 2, device="cuda")
    with pytest.raises(RuntimeError):
        _lib.call("fd_conv3x3_wino_fwd", ctypes.byref(d), t.data_ptr(), t.data_ptr(), None, t.data_ptr(), t.data_ptr(), 0, None,
                  _lib.stream())


@pytest.mark.parametrize("Ci,Co,two_d_min", [(64, 128, None), (64, 128, 1), (256, 256, None), (64, 128, "2p")])
def test_winograd_routing_forward_and_data_gradient_match_torch(Ci, Co, two_d_min, fdtune):
    """FD.conv2d routes eligible 3x3 convs to the Winograd kernel (forward and zero-pad data gradient); values and both
    gradients against torch autograd in float64, and the batched weight re-layout (modes 3 / 4; 5 / 6 for the F(2x2, 3x3) layouts:
    forced onto the small shape, by default on the 256-channel one) against the per-call transform."""
    import fusiondepth_amd.functional as FD
    if two_d_min == "2p":                       # the one-workgroup F(2x2, 3x3) kernel (same U2 layouts, modes 5 / 6) forced onto the small shape
        fdtune.lib(wino_fwd_2dp_min_wgs=1)
    elif two_d_min is not None:
        fdtune.lib(wino_fwd_2d_min=two_d_min)
    torch.manual_seed(5)
    x = torch.randn(2, Ci, 12, 40, device="cuda", requires_grad=True)
    w = torch.nn.Parameter(torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05)
    FD.enable_weight_cache([w])
    y = FD.conv2d(x, w, None, 1, 1)
    gy = torch.randn_like(y)
    gx, gw = torch.autograd.grad(y, [x, w], gy)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yd = F.conv2d(xd, wd, None, 1, 1)
    gxd, gwd = torch.autograd.grad(yd, [xd, wd], gy.double())
    relclose(cpu(y), cpu(yd.float()), "y", rtol=1e-5, arel=3e-6)
    relclose(cpu(gx), cpu(gxd.float()), "dx (Winograd data gradient)", rtol=1e-5, arel=3e-6)
    relclose(cpu(gw), cpu(gwd.float()), "dw", rtol=1e-4, arel=1e-5)
    # change the weights, refresh every cached layout in one batched launch, recompute: must equal a fresh per-call transform
    assert FD.build_weight_plan() >= 2
    with torch.no_grad():
        w.mul_(1.5)
    FD.bump_weights_epoch()
    assert FD.refresh_weight_layouts()
    y2 = FD.conv2d(x, w, None, 1, 1)
    gx2 = torch.autograd.grad(y2, x, gy)[0]
    relclose(cpu(y2), cpu(1.5 * y), "y after batched re-layout", rtol=1e-6, arel=1e-6)
    relclose(cpu(gx2), cpu(1.5 * gx), "dx after batched re-layout", rtol=1e-6, arel=1e-6)


@pytest.mark.parametrize("two_d", [2, 1, 0])
@pytest.mark.parametrize("N,Ci,Co,H,W,mode", [
    (2, 272, 128, 24, 80, "reflect"),   # Refiner decoder (channel-padded concatenations): Cin % 32 == 16, 2-D only with two_d = 2
    (1, 144, 144, 8, 12, "reflect"),
    (2, 112, 32, 10, 16, "reflect"),    # ... with at most 32 output channels (HALFM)
    (1, 80, 64, 4, 6, "zero"),
    (2, 64, 64, 24, 80, "zero"),        # 3 (4) tiles x many pixel slices
    (1, 96, 80, 7, 10, "zero"),         # partial channel tiles on both sides, 35 pairs (< one slice of 64); odd height: 1-D kernel
    (1, 96, 80, 8, 10, "zero"),         # the same with whole 2x2 tiles
    (2, 128, 64, 12, 40, "reflect"),    # decoder ConvBlock: reflect padding
    (3, 64, 128, 5, 6, "reflect"),      # edges everywhere: every pair touches a border
    (3, 64, 128, 6, 6, "reflect"),      # ... every 2x2 tile touches one or two
    (2, 64, 64, 2, 4, "reflect"),       # one tile row: both vertical mirrors in the same tile
    (2, 64, 64, 2, 4, "zero"),
    (5, 64, 64, 48, 160, "zero"),       # layer1 plane: 128 slices, chunks that cross image borders (48 * 80 / 2 tiles per image)
    (3, 512, 512, 6, 20, "zero"),       # layer4: one slice, 8 output channels per finishing workgroup
])
def test_winograd_weight_gradient_vs_float64_reference(N, Ci, Co, H, W, mode, two_d, fdtune):
    """k_wgrad_wino through FD.conv2d's backward, against torch float64 autograd: the transposed F(2x2, 3x3) algorithm (16
    products per 2x2 tile of dY; two_d = 2, the default wherever the height is even; 1: only where Cin % 32 == 0 as in rounds 4-5) and the transposed F(2, 3) algorithm per
    kernel row (4 products per pixel pair and row; fd_tuning.wino_wgrad_2d = 0, and odd heights); also accumulation into an existing
    gradient (the trainer's direct-gradient mode)."""
    import fusiondepth_amd.functional as FD
    fdtune.lib(wino_wgrad_2d=two_d)
    torch.manual_seed(Ci + Co)
    x = torch.randn(N, Ci, H, W, device="cuda")
    w = torch.nn.Parameter(torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05)
    y = FD.conv2d(x, w, None, 1, 1, mode)
    gy = torch.randn_like(y)
    gw = torch.autograd.grad(y, w, gy)[0]
    xd = F.pad(x.double(), (1, 1, 1, 1), mode="reflect" if mode == "reflect" else "constant")
    wd = w.detach().double().requires_grad_(True)
    gwd = torch.autograd.grad(F.conv2d(xd, wd), wd, gy.double())[0]
    relclose(cpu(gw), cpu(gwd.float()), "dw (Winograd weight gradient)", rtol=1e-5, arel=3e-6)
    # direct accumulation into a pre-existing .grad (fd_conv2d_bwd_weight accumulate = 1)
    w.grad = torch.ones_like(w)
    FD.enable_direct_grad([w])
    y2 = FD.conv2d(x, w, None, 1, 1, mode)
    y2.backward(gy)
    relclose(cpu(w.grad), cpu(gwd.float() + 1.0), "accumulated dw", rtol=1e-5, arel=3e-6)


@pytest.mark.parametrize("mode", ["zero", "reflect"])
@pytest.mark.parametrize("N,Ci,Co,H,W", [(1, 80, 64, 4, 6), (2, 64, 64, 24, 80), (1, 96, 80, 8, 10), (2, 64, 64, 2, 8), (5, 64, 64, 48, 160),
                                         (3, 512, 512, 6, 20), (4, 128, 128, 24, 80), (6, 256, 256, 12, 40), (2, 112, 96, 10, 16), (2, 272, 128, 24, 80)])
def test_winograd_weight_gradient_split_precision_vs_float64(N, Ci, Co, H, W, mode, fdtune):
    """k_wgrad_wino_limb (fd_tuning.wino_wgrad_limb: the transposed F(2x2, 3x3) weight gradient of zero-padded layers with the horizontal
    transforms + bf16x3 limb split in the loader and a bf16 matrix loop): error against float64 no worse than 2x the f32 kernel's + 1e-7
    relative to the largest entry, and <= 3e-6; accumulation onto an existing gradient; the route is really taken (log)."""
    import conftest
    import fusiondepth_amd.functional as FD
    torch.manual_seed(Ci + Co)
    x = torch.randn(N, Ci, H, W, device="cuda").relu_()
    w = torch.nn.Parameter(torch.randn(Co, Ci, 3, 3, device="cuda") * 0.05)
    gy = torch.randn(N, Co, H, W, device="cuda")
    xd = F.pad(x.double(), (1, 1, 1, 1), mode="reflect" if mode == "reflect" else "constant")
    wd = w.detach().double().requires_grad_(True)
    gwd = torch.autograd.grad(F.conv2d(xd, wd), wd, gy.double())[0]
    err = {}
    for limb in (0, 2):                      # 2: zero- and reflect-padded layers
        fdtune.lib(wino_wgrad_limb=limb)
        y = FD.conv2d(x, w, None, 1, 1, mode)
        gw = torch.autograd.grad(y, w, gy)[0]
        err[limb] = float((gw.double() - gwd).abs().max() / gwd.abs().max())
    bound = max(3e-6, 2 * err[0] + 1e-7)
    conftest.report("Winograd weight gradient, limb matrix loop b%d %d->%d @%dx%d %s: max |err| / max |ref| vs float64" % (N, Ci, Co, H, W, mode), err[2], bound,
                    "(f32 kernel %.1e)" % err[0])
    assert err[2] <= bound, err
    w.grad = torch.ones_like(w)
    FD.enable_direct_grad([w])
    FD.conv2d(x, w, None, 1, 1, mode).backward(gy)
    relclose(cpu(w.grad), cpu(gwd.float() + 1.0), "accumulated dw (limb)", rtol=1e-5, arel=3e-6)


@pytest.mark.parametrize("N,Ci,Co,H,W,mode", [(12, 256, 256, 12, 40, "zero"), (6, 512, 512, 6, 20, "zero"), (2, 128, 256, 24, 80, "reflect"), (2, 256, 128, 24, 80, "reflect"),
                                              (1, 272, 272, 24, 80, "reflect"), (3, 256, 320, 4, 8, "zero"), (2, 512, 256, 12, 40, "reflect")])
def test_winograd_slab_kernel_split_precision_vs_float64(N, Ci, Co, H, W, mode, FD, fdtune):
    """k_conv_wino2d_limb (fd_tuning.wino_fwd_limb: the F(2x2, 3x3) slab kernel of the deep layers with pre-split weights, a
    once-per-workgroup transform + limb-split stage and a bf16 matrix loop), forward (+ bias, ELU) and data gradient (the same kernel on dY
    with the flipped kernel's image), zero and reflect padding, channel splits, partial channel tiles: error against float64 no worse than
    2x the f32 kernels' + 1e-7 relative to the largest entry and <= 3e-6 / 1e-5; the route is taken (log)."""
    import conftest
    torch.manual_seed(Ci * 3 + Co)
    x = torch.randn(N, Ci, H, W, device="cuda", requires_grad=True)
    w = (torch.randn(Co, Ci, 3, 3, device="cuda") * 0.03).requires_grad_(True)
    b = torch.randn(Co, device="cuda")
    cot = torch.randn(N, Co, H, W, device="cuda")
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double()
    xp = F.pad(xd, (1, 1, 1, 1), mode="reflect" if mode == "reflect" else "constant")
    yd = F.elu(F.conv2d(xp, wd, b.double()))
    gxd, = torch.autograd.grad((yd * cot.double()).sum(), [xd])
    err = {}
    for limb in (0, 1):
        fdtune.lib(wino_fwd_limb=limb)
        y = FD.conv2d(x, w, b, 1, 1, mode, "elu")
        gx, = torch.autograd.grad((y * cot).sum(), [x])
        err[limb] = (float((y.detach().double() - yd.detach()).abs().max() / yd.detach().abs().max()), float((gx.double() - gxd).abs().max() / gxd.abs().max()))
    for k, name, floor in ((0, "forward", 3e-6), (1, "data gradient", 1e-5)):
        bound = max(floor, 2 * err[0][k] + 1e-7)
        conftest.report("Winograd slab kernel, limb matrix loop b%d %d->%d @%dx%d %s %s: max |err| / max |ref| vs float64" % (N, Ci, Co, H, W, mode, name),
                        err[1][k], bound, "(f32 kernel %.1e)" % err[0][k])
        assert err[1][k] <= bound, (name, err)


def test_interleaved_encoders_equal_sequential_passes():
    """networks.interleaved_forward (four encoders advanced block by block in turns, each on its own stream) returns exactly what
    four sequential forward calls return, and autograd through it gives the same parameter gradients."""
    from fusiondepth_amd import networks
    import fusiondepth_amd.functional as FD
    torch.manual_seed(3)
    specs = [dict(num_input_images=2), dict(num_input_images=2, beam_encoder=True), dict(beam_encoder=True), dict()]
    encs = [networks.ResnetEncoder(18, False, **kw).cuda().train() for kw in specs]
    xs = [torch.rand(2, e.encoder.conv1.weight.shape[1], 64, 96, device="cuda") for e in encs]
    seq = []
    for e, x in zip(encs, xs):
        saved = {n: b.clone() for n, b in e.named_buffers()}
        feats = e(x)
        loss = sum(f.square().mean() for f in feats)
        grads = torch.autograd.grad(loss, list(e.parameters()), allow_unused=True)
        seq.append(([f.detach().clone() for f in feats], grads))
        with torch.no_grad():
            for n, b in e.named_buffers():
                b.copy_(saved[n])                     # running statistics back to the start for the second pass
    streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream(), None]
    for st in streams[:3]:
        st.wait_stream(torch.cuda.current_stream())
    outs = networks.interleaved_forward([(e, x, st, 1) for e, x, st in zip(encs, xs, streams)])
    for st in streams[:3]:
        torch.cuda.current_stream().wait_stream(st)
    for (feats_s, grads_s), feats_i, e in zip(seq, outs, encs):
        for a, b in zip(feats_s, feats_i):
            assert torch.equal(a, b.detach())
        loss = sum(f.square().mean() for f in feats_i)
        grads_i = torch.autograd.grad(loss, list(e.parameters()), allow_unused=True)
        torch.cuda.synchronize()
        for ga, gb in zip(grads_s, grads_i):
            assert (ga is None) == (gb is None)
            if ga is not None:
                assert torch.equal(ga, gb)


# ---- act' applied where the gradient is produced (fd_upcat_bwd_act / fd_conv2d_bwd_data_inact had no direct tests) --
_TORCH_ACT = {"relu": torch.relu, "elu": F.elu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}


@pytest.mark.parametrize("act", ["relu", "elu", "sigmoid", "tanh"])
@pytest.mark.parametrize("h,w", [(6, 8), (5, 7), (12, 40)])
def test_upcat_bwd_with_fused_activation_gradient(FD, act, h, w):
    """upsample_concat(a_act=...) after a conv2d(grad_preact=True): the gradient that reaches the convolution's input and weight is
    the one torch autograd gives for  cat([up2(act(conv(x))), skip])  - every activation id, even (vector path) and odd (scalar
    path) widths."""
    rng = np.random.RandomState(21)
    mk = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32))
    N, Cin, Ca, Cs = 2, 16, 16, 5
    x, wgt, bias, skip = mk(N, Cin, h, w), mk(Ca, Cin, 3, 3) * 0.2, mk(Ca) * 0.1, mk(N, Cs, 2 * h, 2 * w)
    xo, wo, bo, so = (t.clone().requires_grad_(True) for t in (x, wgt, bias, skip))
    a = _TORCH_ACT[act](F.conv2d(F.pad(xo, (1, 1, 1, 1), mode="reflect"), wo, bo))
    yo = torch.cat([OL.upsample(a), so], 1)
    cot = mk(*yo.shape)
    want = torch.autograd.grad((yo * cot).sum(), [xo, wo, bo, so])
    for fused in (True, False):
        xg, wg, bg, sg = (dev(t).requires_grad_(True) for t in (x, wgt, bias, skip))
        ag = FD.conv2d(xg, wg, bg, 1, 1, "reflect", act, grad_preact=fused)
        yg = FD.upsample_concat(ag, sg, a_act=act if fused else "none")
        got = torch.autograd.grad((yg * dev(cot)).sum(), [xg, wg, bg, sg])
        relclose(cpu(yg), cpu(yo), "forward", rtol=2e-5, arel=2e-5)
        for name, g1, g2 in zip(("gx", "gw", "gb", "gskip"), got, want):
            relclose(cpu(g1), cpu(g2), "%s (%s, fused=%s)" % (name, act, fused), rtol=5e-5, arel=5e-5)


@pytest.mark.parametrize("act", ["relu", "elu", "sigmoid", "tanh"])
@pytest.mark.parametrize("c1", [1, 0])
def test_data_gradient_with_input_activation(FD, fdtune, act, c1):
    """conv2d(in_act=...) as the single consumer of a conv2d(grad_preact=True): the one-output-channel stencil route (dispconv) and,
    with it switched off (fd_tuning.conv_c1 = 0), the fallback that runs the plain data gradient followed by an in-place act'."""
    fdtune.lib(conv_c1=c1)
    rng = np.random.RandomState(22)
    mk = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32))
    N, Cin, Cm, h, w = 2, 16, 16, 12, 20
    x, w1, b1, w2, b2 = mk(N, Cin, h, w), mk(Cm, Cin, 3, 3) * 0.2, mk(Cm) * 0.1, mk(1, Cm, 3, 3) * 0.2, mk(1) * 0.1
    leaves_o = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    pad = lambda t: F.pad(t, (1, 1, 1, 1), mode="reflect")
    mid = _TORCH_ACT[act](F.conv2d(pad(leaves_o[0]), leaves_o[1], leaves_o[2]))
    yo = torch.sigmoid(F.conv2d(pad(mid), leaves_o[3], leaves_o[4]))
    cot = mk(*yo.shape)
    want = torch.autograd.grad((yo * cot).sum(), leaves_o)
    for fused in (True, False):
        lg = [dev(t).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
        m = FD.conv2d(lg[0], lg[1], lg[2], 1, 1, "reflect", act, grad_preact=fused)
        yg = FD.conv2d(m, lg[3], lg[4], 1, 1, "reflect", "sigmoid", in_act=act if fused else "none")
        got = torch.autograd.grad((yg * dev(cot)).sum(), lg)
        relclose(cpu(yg), cpu(yo), "forward", rtol=2e-5, arel=2e-5)
        for name, g1, g2 in zip(("gx", "gw1", "gb1", "gw2", "gb2"), got, want):
            relclose(cpu(g1), cpu(g2), "%s (%s, conv_c1=%d, fused=%s)" % (name, act, c1, fused), rtol=5e-5, arel=5e-5)


def test_input_activation_contract_is_enforced(FD):
    """a layer built with in_act whose fused data gradient cannot be taken (a second gradient arriving at its input,
    as through conv2d_tap) must raise instead of silently dropping act'."""
    rng = np.random.RandomState(23)
    mk = lambda *s: torch.from_numpy(rng.randn(*s).astype(np.float32))
    x, wgt = dev(mk(1, 16, 8, 8)), dev(mk(16, 16, 3, 3))
    c = FD._Part()
    _, y, _ = FD._conv_forward(c, x, wgt, None, 1, 1, 0, 0, False)
    c.in_act = FD.ACT["elu"]
    with pytest.raises(RuntimeError, match="in_act"):
        FD._conv_backward(c, torch.ones_like(y), gx_add=torch.ones_like(x))
    c.in_act = 0
    gx, gw, _ = FD._conv_backward(c, torch.ones_like(y), gx_add=torch.ones_like(x))
    assert gx.shape == x.shape and gw.shape == wgt.shape


def _seeded_bn(C, rng_seed):
    bn = gin.fill_params(torch.nn.BatchNorm2d(C), rng_seed).cuda()
    bn.train()
    return bn


@pytest.mark.parametrize("N,C,H,W,groups", [(4, 64, 96, 320, 2), (2, 8, 7, 9, 1), (3, 5, 10, 13, 1), (6, 16, 48, 160, 3), (2, 4, 9, 12, 1), (2, 3, 6, 4, 2)])
@pytest.mark.parametrize("want_feat", [True, False])
def test_fused_stem_tail_equals_batchnorm_relu_maxpool(FD, N, C, H, W, groups, want_feat):
    """fd_bn_relu_maxpool_fwd / _bwd (one pass each way over the stem's output, features[0] written only on request) against the
    separate BatchNorm(+ReLU) and max-pool calls: pooled values, argmax routing, features[0], running statistics and saved statistics
    identical; input / weight / bias gradients equal up to the summation order of the per-channel reductions.  Even and odd plane
    sizes, grouped statistics, with and without a second gradient arriving at features[0]."""
    rng = np.random.RandomState(N * 100 + H)
    x = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32) * 1.5 + 0.3).cuda()
    mk_bn = lambda: _seeded_bn(C, rng_seed=5)
    cot_p = torch.from_numpy(rng.randn(N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1).astype(np.float32)).cuda()
    cot_f = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32)).cuda()
    res = {}
    for fused in (True, False):
        bn = mk_bn()
        xg = x.clone().requires_grad_(True)
        with FD.bn_groups(groups):
            if fused:
                f0, p = FD.bn_relu_maxpool(xg, bn, want_feature=want_feat)
            else:
                f0 = FD.batch_norm(xg, bn, relu=True)
                p = FD.max_pool3x3s2(f0)
        loss = (p * cot_p).sum()
        if want_feat:
            loss = loss + (f0 * cot_f).sum()
        gx, gw, gb = torch.autograd.grad(loss, [xg, bn.weight, bn.bias])
        res[fused] = (p.detach(), f0.detach() if want_feat else None, gx, gw, gb, bn.running_mean.clone(), bn.running_var.clone(),
                      int(bn.num_batches_tracked))
    a, b = res[True], res[False]
    assert a[7] == b[7] == groups
    relclose(cpu(a[0]), cpu(b[0]), "pooled", rtol=1e-6, arel=1e-6)
    if want_feat:
        relclose(cpu(a[1]), cpu(b[1]), "features[0]", rtol=1e-6, arel=1e-6)
    else:
        assert a[1] is None
    for i, name in ((5, "running_mean"), (6, "running_var")):      # bit-equal on the large planes; tiny ones take the unfused path's
        relclose(cpu(a[i]), cpu(b[i]), name, rtol=1e-6, arel=1e-6)  # small-plane kernel (statistics shifted by another sample)
    for i, name in ((2, "gx"), (3, "gweight"), (4, "gbias")):
        relclose(cpu(a[i]), cpu(b[i]), name, rtol=2e-5, arel=2e-5)


def test_stack_normalize_is_cat_plus_input_normalisation(FD):
    """fd_stack_normalize: the pose networks' input - torch.cat of frame pairs along the channels, the pairs of all source frames and
    micro-batches along the batch axis, then (x - 0.45) / 0.225 - in one launch, bit for bit (a true division, like the reference)."""
    rng = np.random.RandomState(31)
    G, Bg, C, H, W = 2, 3, 3, 8, 12
    frames = {f: torch.from_numpy(rng.rand(G * Bg, C, H, W).astype(np.float32)).cuda() for f in (-1, 0, 1)}
    orders = [(-1, 0), (0, 1)]
    pieces, want = [], []
    for g in range(G):
        for k, o in enumerate(orders):
            for j, i in enumerate(o):
                pieces.append((frames[i][g * Bg:(g + 1) * Bg], (g * len(orders) + k) * Bg, j * C))
            want.append(torch.cat([frames[i][g * Bg:(g + 1) * Bg] for i in o], 1))
    want = torch.cat(want, 0)
    got = FD.stack_normalize(pieces, G * len(orders) * Bg, 2 * C)
    assert torch.equal(got.cpu(), (want.cpu() - 0.45) / 0.225)          # on the CPU: ATen's GPU kernel multiplies by the reciprocal instead
    assert torch.equal(FD.stack_normalize(pieces, G * len(orders) * Bg, 2 * C, normalize=False), want)
    many = [(frames[0][:1], i, 0) for i in range(20)]                      # more than 16 pieces: several launches
    assert torch.equal(FD.stack_normalize(many, 20, C, normalize=False), frames[0][:1].repeat(20, 1, 1, 1))


@pytest.mark.parametrize("N,C,H,W,groups,res", [(8, 256, 12, 40, 2, True), (12, 512, 6, 20, 2, False), (4, 256, 12, 40, 1, True), (24, 512, 6, 20, 4, True)])
def test_conv_bn_with_the_slab_reduction_inside_the_batchnorm_is_bit_identical(FD, fdtune, N, C, H, W, groups, res):
    """fd_conv2d_fwd_bn (round 5): for the deep ResNet layers the F(2x2, 3x3) slab reduction runs inside the small-plane BatchNorm kernel
    instead of as k_wino2d_finish.  Same additions in the same order: the block's output, the running statistics and every gradient of
    conv_bn must be BIT-identical with the switch on and off."""
    from fusiondepth_amd import _lib
    import ctypes
    rng = np.random.RandomState(C + N)
    x = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32)).cuda()
    w = torch.from_numpy((rng.randn(C, C, 3, 3) * 0.02).astype(np.float32)).cuda()
    r = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32)).cuda() if res else None
    cot = torch.from_numpy(rng.randn(N, C, H, W).astype(np.float32)).cuda()
    d = _lib.ConvDesc(N, C, H, W, C, 3, 3, 1, 1, 0, 0, 0)
    assert _lib.query("fd_conv2d_fwd_bn_ok", ctypes.byref(d), groups) == 1
    out = {}
    for on in (True, False):
        fdtune.host(fused_finish_bn=on)
        bn = _seeded_bn(C, rng_seed=3)
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        rg = r.clone().requires_grad_(True) if res else None
        with FD.bn_groups(groups):
            y = FD.conv_bn(xg, wg, bn, stride=1, pad=1, residual=rg, relu=True)
        grads = torch.autograd.grad((y * cot).sum(), [xg, wg, bn.weight, bn.bias] + ([rg] if res else []))
        out[on] = [y.detach(), bn.running_mean.clone(), bn.running_var.clone()] + [g.detach() for g in grads]
    for a, b in zip(out[True], out[False]):
        assert torch.equal(a, b), float((a - b).abs().max())
