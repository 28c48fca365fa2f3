"""GPU parity tests (run with ``-m gpu`` on the MI355X box): HIP loss-path kernels, called through the
C ABI, against the CPU oracle on identical seeded inputs and against the committed golden vectors.
Tolerance: the north-star 1e-4 relative bound on loss/depth tensors (tighter where cheap)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import inputs as gin
from conftest import assert_close, assert_mostly_close
from oracle import layers as OL
from oracle import scatter as OS
from oracle import trainer as OT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def FD():
    from fusiondepth_amd import functional
    return functional


def dev(t):
    return t.detach().clone().cuda()


def cpu(t):
    return t.detach().cpu().numpy()


def grads(loss, leaves):
    return [cpu(g) for g in torch.autograd.grad(loss, leaves)]


# ------------------------------------------------------------------------------------------------
def test_disp_to_depth_and_pose_matrix(FD, golden):
    g = golden("layers_b2_32x64")
    B, H, W = 2, 32, 64
    inp, rng = gin.batch_inputs(101, B, H, W)
    disp = gin.disp_pyramid(rng, B, H, W)[("disp", 0)]
    d = dev(disp).requires_grad_(True)
    sd, depth = FD.disp_to_depth(d, 0.1, 100.0)
    assert_close(cpu(sd), g["d2d_scaled"], rtol=1e-6, atol=0, what="scaled disp")
    assert_close(cpu(depth), g["d2d_depth"], rtol=1e-6, atol=0, what="depth")
    cot = torch.from_numpy(np.random.RandomState(1).randn(2, B, 1, H, W).astype(np.float32))
    d_o = disp.clone().requires_grad_(True)
    so, do = OL.disp_to_depth(d_o, 0.1, 100.0)
    want = grads((so * cot[0]).sum() + (do * cot[1]).sum(), [d_o])[0]
    got = grads((sd * dev(cot[0])).sum() + (depth * dev(cot[1])).sum(), [d])[0]
    assert_close(got, want, rtol=1e-5, atol=1e-6 * np.abs(want).max(), what="disp_to_depth grad")

    aa, tr = gin.small_poses(rng, B)
    cot_T = torch.from_numpy(g["T_cot"])
    for inv in (False, True):
        tag = "inv" if inv else "fwd"
        a, t = dev(aa).requires_grad_(True), dev(tr).requires_grad_(True)
        M = FD.transformation_from_parameters(a, t, invert=inv)
        ga, gt = grads((M * dev(cot_T)).sum(), [a, t])
        assert_close(cpu(M), g["T_" + tag], rtol=1e-5, atol=1e-7, what="T " + tag)
        assert_close(ga, g["T_%s_gaa" % tag], rtol=1e-4, atol=1e-5, what="g axisangle " + tag)
        assert_close(gt, g["T_%s_gtr" % tag], rtol=1e-4, atol=1e-6, what="g translation " + tag)
    # zero rotation: norm() has a 0 sub-gradient at the origin (must not produce NaN)
    z = torch.zeros(B, 1, 3).cuda().requires_grad_(True)
    M = FD.transformation_from_parameters(z, dev(tr), invert=False)
    gz = torch.autograd.grad(M.sum(), z)[0]
    assert torch.isfinite(gz).all() and torch.isfinite(M).all()


@pytest.mark.parametrize("G,nf,Bq,npred", [(2, 2, 3, 2), (1, 2, 6, 2), (2, 1, 4, 1), (3, 4, 2, 2)])
def test_pose_head_equals_slices_plus_pose_matrices(FD, G, nf, Bq, npred):
    """fd_pose_head_fwd / _bwd (one launch each way for every frame pair and accumulated micro-batch of the stacked pose network) against
    the path it replaces - trainer.py:338-360 as slices, concatenations and one fd_pose_matrix_* call per frame pair: bit for bit, forward
    and backward, including a frame pair whose matrix receives no gradient."""
    torch.manual_seed(G * 100 + nf * 10 + Bq)
    pose = (torch.randn(G * nf * Bq, 6 * npred, device="cuda") * 0.05)
    inverts = [k % 2 == 0 for k in range(nf)]
    cots = [torch.randn(G * Bq, 4, 4, device="cuda") for _ in range(nf)]
    unused = nf - 1 if nf > 1 else None                       # this pair's matrix takes no part in the loss
    p1 = pose.clone().requires_grad_(True)
    heads = FD.pose_head(p1, G, nf, Bq, inverts)
    loss = sum((heads[k][0] * cots[k]).sum() for k in range(nf) if k != unused)
    g1 = torch.autograd.grad(loss, p1)[0]
    p2 = pose.clone().requires_grad_(True)
    out = p2.view(-1, npred, 1, 6)
    aa_all, tr_all = out[..., :3], out[..., 3:]
    loss2 = 0
    for k in range(nf):
        sl = [slice((g * nf + k) * Bq, (g * nf + k + 1) * Bq) for g in range(G)]
        aa = torch.cat([aa_all[q] for q in sl], 0)
        tr = torch.cat([tr_all[q] for q in sl], 0)
        T = FD.transformation_from_parameters(aa[:, 0], tr[:, 0], invert=inverts[k])
        assert torch.equal(heads[k][0], T), "T of pair %d" % k
        assert torch.equal(heads[k][1], aa) and torch.equal(heads[k][2], tr)
        assert not heads[k][1].requires_grad and not heads[k][2].requires_grad
        if k != unused:
            loss2 = loss2 + (T * cots[k]).sum()
    g2 = torch.autograd.grad(loss2, p2)[0]
    assert torch.equal(g1, g2), "%d gradient entries differ" % int((g1 != g2).sum())
    with pytest.raises(ValueError):
        FD.pose_head(pose[:-1], G, nf, Bq, inverts)


def test_backproject_project_against_golden(FD, golden):
    g = golden("layers_b2_32x64")
    B, H, W = 2, 32, 64
    inp, rng = gin.batch_inputs(101, B, H, W)
    K, inv_K = dev(inp[("K", 0)]), dev(inp[("inv_K", 0)])
    depth = torch.from_numpy(g["d2d_depth"]).cuda().requires_grad_(True)
    T = torch.from_numpy(g["pj_T"]).cuda().requires_grad_(True)
    pts = FD.backproject_depth(depth, inv_K)
    grid = FD.project_3d(pts, K, T, H, W)
    gd, gT = grads((grid * torch.from_numpy(g["pj_cot"]).cuda()).sum(), [depth, T])
    assert_close(cpu(pts), g["bp_points"], rtol=1e-5, atol=1e-6, what="backproject")
    assert_close(cpu(grid), g["pj_grid"], rtol=1e-4, atol=1e-5, what="project grid")
    assert_close(gd, g["pj_gdepth"], rtol=1e-4, atol=1e-5, what="g depth")
    assert_close(gT, g["pj_gT"], rtol=1e-4, atol=1e-3, what="g T")
    assert_close(cpu(FD.cat_xy(depth, inv_K)), g["catxy"], rtol=1e-5, atol=1e-6, what="Cat_xy")


@pytest.mark.parametrize("hin,win,hout,wout", [(24, 80, 192, 640), (48, 160, 192, 640), (96, 320, 192, 640),
                                               (192, 640, 192, 640), (8, 12, 64, 96), (5, 7, 10, 14)])
def test_bilinear_upsample(FD, hin, win, hout, wout):
    rng = np.random.RandomState(3)
    x = torch.from_numpy(rng.rand(2, 3, hin, win).astype(np.float32))
    cot = torch.from_numpy(rng.randn(2, 3, hout, wout).astype(np.float32))
    xo = x.clone().requires_grad_(True)
    yo = F.interpolate(xo, [hout, wout], mode="bilinear", align_corners=False)
    want_g = grads((yo * cot).sum(), [xo])[0]
    xg = dev(x).requires_grad_(True)
    yg = FD.bilinear_upsample(xg, (hout, wout))
    got_g = grads((yg * dev(cot)).sum(), [xg])[0]
    assert_close(cpu(yg), cpu(yo), rtol=1e-5, atol=1e-6, what="bilinear fwd")
    assert_close(got_g, want_g, rtol=1e-4, atol=1e-5, what="bilinear bwd")


@pytest.mark.parametrize("B,C,H,W", [(2, 3, 32, 64), (1, 3, 20, 70), (1, 2, 192, 640), (1, 1, 4, 4), (1, 1, 17, 129)])
def test_ssim_fwd_bwd(FD, golden, B, C, H, W):
    rng = np.random.RandomState(5)
    if (B, C, H, W) == (2, 3, 32, 64):
        g = golden("layers_b2_32x64")
        inp, _ = gin.batch_inputs(101, B, H, W)
        x, y, cot = inp[("color", 0, 0)], inp[("color", 1, 0)], torch.from_numpy(g["ssim_cot"])
    else:
        x = torch.from_numpy(gin.smooth_image(rng, B, C, H, W))
        y = torch.from_numpy(np.clip(x.numpy() + 0.05 * rng.randn(B, C, H, W), 0, 1).astype(np.float32))
        cot = torch.from_numpy(rng.rand(B, C, H, W).astype(np.float32))
    xo, yo = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    so = OL.ssim(xo, yo)
    want = grads((so * cot).sum(), [xo, yo])
    xg, yg = dev(x).requires_grad_(True), dev(y).requires_grad_(True)
    sg = FD.ssim(xg, yg)
    got = grads((sg * dev(cot)).sum(), [xg, yg])
    assert_close(cpu(sg), cpu(so), rtol=1e-4, atol=2e-5, what="ssim fwd")
    if (B, C, H, W) == (2, 3, 32, 64):
        assert_close(cpu(sg), g["ssim"], rtol=1e-4, atol=2e-5, what="ssim fwd vs golden")
        assert_close(got[0], g["ssim_gx"], rtol=1e-3, atol=2e-4, what="ssim gx vs golden")
    scale = np.abs(want[0]).max()
    assert_close(got[0], want[0], rtol=1e-3, atol=1e-4 * scale, what="ssim gx")
    assert_close(got[1], want[1], rtol=1e-3, atol=1e-4 * scale, what="ssim gy")


@pytest.mark.parametrize("use_ssim", [True, False])
def test_reprojection_loss_map(FD, use_ssim):
    B, H, W = 2, 40, 100
    inp, rng = gin.batch_inputs(77, B, H, W)
    opt = OT.default_opt(no_ssim=not use_ssim)
    want = OT.reprojection_loss(opt, inp[("color", -1, 0)], inp[("color", 0, 0)])
    got = FD.reprojection_loss_map(dev(inp[("color", -1, 0)]), dev(inp[("color", 0, 0)]), use_ssim)
    assert_close(cpu(got), cpu(want), rtol=1e-4, atol=2e-5, what="reprojection loss map")


@pytest.mark.parametrize("B,H,W", [(2, 32, 64), (2, 24, 80), (1, 192, 640), (3, 5, 9)])
def test_smooth_loss(FD, golden, B, H, W):
    inp, rng = gin.batch_inputs(101 if (H, W) == (32, 64) else 9, B, H, W, num_scales=4 if H % 8 == 0 else 1)
    disp = gin.disp_pyramid(rng, B, H, W, num_scales=4 if H % 8 == 0 else 1)[("disp", 0)]
    img = inp[("color", 0, 0)]
    for normalize in (False, True):
        do = disp.clone().requires_grad_(True)
        n = do / (do.mean(2, True).mean(3, True) + 1e-7) if normalize else do
        lo = OL.get_smooth_loss(n, img)
        want_g = grads(lo * 1.7, [do])[0]
        dg = dev(disp).requires_grad_(True)
        lg = FD.normalized_smooth_loss(dg, dev(img)) if normalize else FD.get_smooth_loss(dg, dev(img))
        got_g = grads(lg * 1.7, [dg])[0]
        assert_close(cpu(lg), cpu(lo), rtol=1e-5, atol=1e-8, what="smooth loss normalize=%s" % normalize)
        assert_close(got_g, want_g, rtol=1e-3, atol=1e-5 * np.abs(want_g).max(), what="smooth grad normalize=%s" % normalize)
        if (B, H, W) == (2, 32, 64) and not normalize:
            g = golden("layers_b2_32x64")
            assert_close(cpu(lg), g["smooth"], rtol=1e-5, atol=1e-8, what="smooth vs golden")


def test_scatter_2channel_bit_exact(FD, golden):
    g = golden("scatter_192x640")
    beams = np.stack([g["beam%d" % i] for i in range(3)])[:, None]
    out = cpu(FD.scatter_2channel(torch.from_numpy(beams).cuda()))
    for i in range(3):
        assert np.array_equal(out[i, 1], g["conf%d" % i]), "confidence %d differs from the reference" % i
        assert np.array_equal(out[i, 0], g["depth%d" % i]), "depth %d: max diff %g" % (i, np.abs(out[i, 0] - g["depth%d" % i]).max())
    # random dense/sparse maps vs the sequential C oracle, incl. empty and ROI-edge cases
    rng = np.random.RandomState(11)
    maps = [np.zeros((192, 640), np.float32)]
    for dens in (0.002, 0.05, 0.5):
        m = (rng.rand(192, 640) < dens) * rng.uniform(0.01, 0.8, size=(192, 640))
        maps.append(m.astype(np.float32))
    got = cpu(FD.scatter_2channel(torch.from_numpy(np.stack(maps)[:, None]).cuda()))
    for i, m in enumerate(maps):
        d, c = OS.scatter_2channel_c(m)
        assert np.array_equal(got[i, 0], d) and np.array_equal(got[i, 1], c), "random map %d" % i
    with pytest.raises(RuntimeError):
        FD.scatter_2channel(torch.zeros(1, 1, 192, 640).cuda(), roi=(0, 190, 2, 638))


# ------------------------------------------------------------------------------------------------
def _oracle_photo_terms(opt, inp, disp, Ts, noise):
    """Per scale (to_optimise.mean(), si_loss) + outputs dict, from the oracle (trainer.py:425-589)."""
    outputs = {("disp", s): disp[s] for s in opt.scales}
    for f, T in Ts.items():
        outputs[("cam_T_cam", 0, f)] = T
    OT.generate_images_pred(opt, inp, outputs)
    terms = []
    for s in opt.scales:
        target = inp[("color", 0, 0)]
        reproj = torch.cat([OT.reprojection_loss(opt, outputs[("color", f, s)], target) for f in opt.frame_ids[1:]], 1)
        if opt.avg_reprojection:
            reproj = reproj.mean(1, keepdim=True)
        if not opt.disable_automasking:
            ident = torch.cat([OT.reprojection_loss(opt, inp[("color", f, 0)], target) for f in opt.frame_ids[1:]], 1)
            if opt.avg_reprojection:
                ident = ident.mean(1, keepdim=True)
            comb = torch.cat((ident + noise[s] * 0.00001, reproj), 1)
        else:
            comb = reproj
        mn, idx = (comb[:, 0], torch.zeros_like(comb[:, 0], dtype=torch.long)) if comb.shape[1] == 1 else torch.min(comb, dim=1)
        terms.append((mn.mean(), OT.si_log_loss(opt, disp[s], inp["4beam"]), idx))
    return terms, outputs


def _hip_photo_terms(FD, opt, inp, disp, Ts, noise, materialize=True, groups=1):
    po = FD.PhotoOptions(opt.min_depth, opt.max_depth, opt.no_ssim, opt.avg_reprojection, opt.gdc_loss_threshold, opt.si_var)
    fids = opt.frame_ids[1:]
    target = dev(inp[("color", 0, 0)])
    srcs = [dev(inp[("color", f, 0)]) for f in fids]
    ident = None
    if not opt.disable_automasking:
        ident = torch.cat([FD.reprojection_loss_map(s, target, not opt.no_ssim) for s in srcs], 1)
        if opt.avg_reprojection:
            ident = ident.mean(1, keepdim=True)
    K, inv_K, beam = dev(inp[("K", 0)]), dev(inp[("inv_K", 0)]), dev(inp["4beam"])
    res = []
    for s in opt.scales:
        nz = dev(noise[s]) if ident is not None else None
        res.append(FD.photo_loss(disp[s], [Ts[f] for f in fids], K, inv_K, srcs, target, ident, nz, beam, po, materialize, groups))
    return res


def _photo_case(FD, seed, B, H, W, **over):
    opt = OT.default_opt(height=H, width=W, **over)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp0 = gin.disp_pyramid(rng, B, H, W)
    poses = {f: gin.small_poses(rng, B) for f in (-1, 1)}
    T0 = {f: OL.transformation_from_parameters(*poses[f], invert=(f < 0)) for f in (-1, 1)}
    noise = [torch.from_numpy(np.random.RandomState(1000 + seed + s).randn(B, 1 if opt.avg_reprojection else 2, H, W)
                              .astype(np.float32)) for s in range(4)]
    d_o = {s: disp0[("disp", s)].clone().requires_grad_(True) for s in range(4)}
    T_o = {f: T0[f].clone().requires_grad_(True) for f in T0}
    terms, outs = _oracle_photo_terms(opt, inp, d_o, T_o, noise)
    d_g = {s: dev(disp0[("disp", s)]).requires_grad_(True) for s in range(4)}
    T_g = {f: dev(T0[f]).requires_grad_(True) for f in T0}
    res = _hip_photo_terms(FD, opt, inp, d_g, T_g, noise)
    return opt, terms, outs, res, d_o, T_o, d_g, T_g


@pytest.mark.parametrize("seed,B,H,W,over", [
    (404, 2, 64, 96, {}),
    (31, 2, 48, 200, {}),                       # partial tiles in x, several tiles in y
    (505, 1, 192, 640, {}),                     # full size
    (606, 2, 64, 96, dict(no_ssim=True, disable_automasking=True)),
    (707, 2, 64, 96, dict(avg_reprojection=True)),
    (808, 1, 32, 64, dict(no_ssim=True)),
])
def test_fused_photo_loss_vs_oracle(FD, seed, B, H, W, over):
    opt, terms, outs, res, d_o, T_o, d_g, T_g = _photo_case(FD, seed, B, H, W, **over)
    fids = opt.frame_ids[1:]
    tot_o, tot_g, flips = 0, 0, 0
    for s in range(4):
        photo, si, sel, depth, sample, color = res[s]
        assert_close(cpu(depth), cpu(outs[("depth", 0, s)]), rtol=1e-5, atol=1e-6, what="depth s%d" % s)
        for i, f in enumerate(fids):
            assert_close(cpu(sample[i]), cpu(outs[("sample", f, s)]), rtol=1e-4, atol=2e-5, what="sample f%d s%d" % (f, s))
            assert_close(cpu(color[i]), cpu(outs[("color", f, s)]), rtol=1e-4, atol=2e-5, what="color f%d s%d" % (f, s))
        mism = (cpu(sel).astype(np.int64) != cpu(terms[s][2])).mean()
        flips += int((cpu(sel).astype(np.int64) != cpu(terms[s][2])).sum())
        assert mism <= 2e-4, "argmin differs on %.4f%% of pixels at scale %d" % (100 * mism, s)
        assert_close(cpu(photo), cpu(terms[s][0]), rtol=1e-4, atol=1e-7, what="to_optimise.mean() s%d" % s)
        assert_close(cpu(si), cpu(terms[s][1]), rtol=1e-4, atol=1e-7, what="si_loss s%d" % s)
        w = 1.0 + 0.25 * s          # distinct cotangents per scale / term
        tot_o = tot_o + w * terms[s][0] + (2.0 - 0.3 * s) * terms[s][1]
        tot_g = tot_g + w * photo + (2.0 - 0.3 * s) * si
    want = grads(tot_o, [d_o[s] for s in range(4)] + [T_o[f] for f in fids])
    got = grads(tot_g, [d_g[s] for s in range(4)] + [T_g[f] for f in fids])
    for s in range(4):
        sc = np.abs(want[s]).max()
        assert_mostly_close(got[s], want[s], rtol=2e-3, atol=2e-4 * sc, what="d loss / d disp s%d" % s)
    for i, f in enumerate(fids):
        sc = np.abs(want[4 + i]).max()
        # a pixel whose argmin flips (measured: <=1e-4 of pixels, always within rounding of a tie) changes the
        # pose gradient, a sum over all pixels, by up to a few % of its largest entry; without flips it is ~1e-6
        assert_close(got[4 + i], want[4 + i], rtol=1e-3, atol=(3e-2 if flips else 1e-4) * sc, what="d loss / d T f%d" % f)


@pytest.mark.parametrize("avg,ssim", [(False, True), (True, True), (False, False)])
def test_three_source_frames_and_predictive_mask_vs_oracle(FD, avg, ssim):
    """fd_photo_fwd_ex / fd_photo_bwd_ex: frames -1, +1 and the stereo partner "s" (trainer.py:61-64, 444-447) with the
    predictive-mask weighting of the reprojection losses (trainer.py:530-541): value, d/d disp, d/d pose of all three frames and
    d/d mask against the oracle's autograd."""
    import torch.nn.functional as F
    B, H, W, seed = 2, 64, 96, 909
    opt = OT.default_opt(height=H, width=W, disable_automasking=True, avg_reprojection=avg, no_ssim=not ssim, use_stereo=True)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    shifted = torch.roll(inp[("color", 1, 0)], 2, dims=3) * 0.97 + 0.01
    for s in range(4):
        inp[("color", "s", s)] = shifted if s == 0 else F.avg_pool2d(shifted, 2 ** s)
    disp0 = gin.disp_pyramid(rng, B, H, W)
    fids = [-1, 1, "s"]
    poses = {f: gin.small_poses(rng, B) for f in fids}
    T0 = {f: OL.transformation_from_parameters(*poses[f], invert=(f == -1)) for f in fids}
    masks0 = {s: torch.from_numpy(rng.uniform(0.15, 0.95, size=(B, 3, H >> s, W >> s)).astype(np.float32)) for s in range(4)}
    # oracle
    d_o = {s: disp0[("disp", s)].clone().requires_grad_(True) for s in range(4)}
    T_o = {f: T0[f].clone().requires_grad_(True) for f in fids}
    m_o = {s: masks0[s].clone().requires_grad_(True) for s in range(4)}
    oin = dict(inp)
    oin["stereo_T"] = T_o["s"]
    outs = {("disp", s): d_o[s] for s in range(4)}
    for f in (-1, 1):
        outs[("cam_T_cam", 0, f)] = T_o[f]
    OT.generate_images_pred(opt, oin, outs)
    tot_o, vals_o, idx_o = 0, [], []
    for s in range(4):
        reproj = torch.cat([OT.reprojection_loss(opt, outs[("color", f, s)], inp[("color", 0, 0)]) for f in fids], 1)
        reproj = reproj * F.interpolate(m_o[s], [H, W], mode="bilinear", align_corners=False)
        if avg:
            v, idx = reproj.mean(1), None
        else:
            v, idx = torch.min(reproj, dim=1)
        vals_o.append(v.mean()); idx_o.append(idx)
        tot_o = tot_o + (1.0 + 0.25 * s) * vals_o[-1]
    leaves_o = [d_o[s] for s in range(4)] + [T_o[f] for f in fids] + [m_o[s] for s in range(4)]
    want = grads(tot_o, leaves_o)
    # HIP
    po = FD.PhotoOptions(opt.min_depth, opt.max_depth, opt.no_ssim, opt.avg_reprojection, opt.gdc_loss_threshold, opt.si_var)
    d_g = {s: dev(disp0[("disp", s)]).requires_grad_(True) for s in range(4)}
    T_g = {f: dev(T0[f]).requires_grad_(True) for f in fids}
    m_g = {s: dev(masks0[s]).requires_grad_(True) for s in range(4)}
    target, srcs = dev(inp[("color", 0, 0)]), [dev(inp[("color", f, 0)]) for f in fids]
    K, inv_K = dev(inp[("K", 0)]), dev(inp[("inv_K", 0)])
    tot_g, flips = 0, 0
    for s in range(4):
        mask_up = FD.bilinear_upsample(m_g[s], (H, W)) if s else m_g[s]
        photo, si, sel, *_ = FD.photo_loss(d_g[s], [T_g[f] for f in fids], K, inv_K, srcs, target, None, None, None, po, False, 1,
                                           mask=mask_up)
        assert_close(cpu(photo), cpu(vals_o[s]), rtol=1e-4, atol=1e-7, what="masked 3-frame loss s%d" % s)
        if not avg:
            diff = cpu(sel).astype(np.int64) != cpu(idx_o[s])
            flips += int(diff.sum())
            assert diff.mean() <= 3e-4, "argmin differs on %.4f%% of pixels at scale %d" % (100 * diff.mean(), s)
        tot_g = tot_g + (1.0 + 0.25 * s) * photo
    got = grads(tot_g, [d_g[s] for s in range(4)] + [T_g[f] for f in fids] + [m_g[s] for s in range(4)])
    for s in range(4):
        sc = np.abs(want[s]).max()
        assert_mostly_close(got[s], want[s], rtol=2e-3, atol=2e-4 * sc, what="d loss / d disp s%d (3 frames, mask)" % s)
        sc = np.abs(want[7 + s]).max()
        assert_mostly_close(got[7 + s], want[7 + s], rtol=2e-3, atol=2e-4 * sc, what="d loss / d mask s%d" % s)
    for i, f in enumerate(fids):
        sc = np.abs(want[4 + i]).max()
        assert_close(got[4 + i], want[4 + i], rtol=1e-3, atol=(3e-2 if flips else 1e-4) * sc, what="d loss / d T frame %s" % f)


def _hip_photo_terms_ms(FD, opt, inp, disp, Ts, noise, rows=0, beam_scales=(0, 1, 2, 3), groups=1):
    po = FD.PhotoOptions(opt.min_depth, opt.max_depth, opt.no_ssim, opt.avg_reprojection, opt.gdc_loss_threshold, opt.si_var)
    fids = opt.frame_ids[1:]
    target = dev(inp[("color", 0, 0)])
    srcs = [dev(inp[("color", f, 0)]) for f in fids]
    ident = None
    if not opt.disable_automasking:
        ident = torch.cat([FD.reprojection_loss_map(s, target, not opt.no_ssim) for s in srcs], 1)
    K, inv_K, beam = dev(inp[("K", 0)]), dev(inp[("inv_K", 0)]), dev(inp["4beam"])
    nz = [dev(noise[s]) for s in opt.scales] if ident is not None else None
    return FD.photo_loss_ms([disp[s] for s in opt.scales], [Ts[f] for f in fids], K, inv_K, srcs, target, ident, nz, beam,
                            beam_scales, po, groups, rows)


@pytest.mark.parametrize("seed,B,H,W,rows,over", [
    (404, 2, 64, 96, 0, {}),
    (404, 2, 64, 96, 7, {}),                      # ragged strips: 64 = 9 x 7 + 1
    (31, 2, 48, 200, 16, {}),                     # several strips in x (200 = 3 x 60 + 20) and y
    (505, 1, 192, 640, 0, {}),                    # full size
    (909, 2, 32, 184, 0, {}),                     # four owned columns in the last strip
    (606, 2, 64, 96, 0, dict(disable_automasking=True)),
])
def test_multiscale_photo_loss_vs_oracle(FD, seed, B, H, W, rows, over):
    """fd_photo_ms_fwd / fd_photo_ms_bwd (all scales in one launch, gradient produced in the forward pass) against the
    oracle's generate_images_pred + compute_losses (trainer.py:425-589) on the same seeded inputs."""
    opt = OT.default_opt(height=H, width=W, **over)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp0 = gin.disp_pyramid(rng, B, H, W)
    poses = {f: gin.small_poses(rng, B) for f in (-1, 1)}
    T0 = {f: OL.transformation_from_parameters(*poses[f], invert=(f < 0)) for f in (-1, 1)}
    noise = [torch.from_numpy(np.random.RandomState(1000 + seed + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
    d_o = {s: disp0[("disp", s)].clone().requires_grad_(True) for s in range(4)}
    T_o = {f: T0[f].clone().requires_grad_(True) for f in T0}
    terms, outs = _oracle_photo_terms(opt, inp, d_o, T_o, noise)
    d_g = {s: dev(disp0[("disp", s)]).requires_grad_(True) for s in range(4)}
    T_g = {f: dev(T0[f]).requires_grad_(True) for f in T0}
    photo, si, sel = _hip_photo_terms_ms(FD, opt, inp, d_g, T_g, noise, rows)
    fids = opt.frame_ids[1:]
    tot_o, tot_g, flips = 0, 0, 0
    for s in range(4):
        diff = cpu(sel[s]).astype(np.int64) != cpu(terms[s][2])
        flips += int(diff.sum())
        assert diff.mean() <= 3e-4, "argmin differs on %.4f%% of pixels at scale %d" % (100 * diff.mean(), s)
        assert_close(cpu(photo[s]), cpu(terms[s][0]), rtol=1e-4, atol=1e-7, what="to_optimise.mean() s%d" % s)
        assert_close(cpu(si[s]), cpu(terms[s][1]), rtol=1e-4, atol=1e-7, what="si_loss s%d" % s)
        w = 1.0 + 0.25 * s
        tot_o = tot_o + w * terms[s][0] + (2.0 - 0.3 * s) * terms[s][1]
        tot_g = tot_g + w * photo[s] + (2.0 - 0.3 * s) * si[s]
    want = grads(tot_o, [d_o[s] for s in range(4)] + [T_o[f] for f in fids])
    got = grads(tot_g, [d_g[s] for s in range(4)] + [T_g[f] for f in fids])
    for s in range(4):
        sc = np.abs(want[s]).max()
        assert_mostly_close(got[s], want[s], rtol=2e-3, atol=2e-4 * sc, what="d loss / d disp s%d" % s)
        # a pixel whose clamp / |.| / border-clip branch (not only the argmin) sits within rounding of a tie shows up as an
        # isolated O(1) error in the disparity gradient: it moves the pose gradient, a sum over all pixels, like an argmin flip
        flips += int((np.abs(got[s] - want[s]) > 1e-3 * sc).sum())
    for i, f in enumerate(fids):
        sc = np.abs(want[4 + i]).max()
        assert_close(got[4 + i], want[4 + i], rtol=1e-3, atol=(3e-2 if flips else 1e-4) * sc, what="d loss / d T f%d" % f)


def _grad_error_stats(g, g64):
    sc = np.abs(g64).max()
    err = np.abs(g.astype(np.float64) - g64)
    return (err > 2e-4 * sc + 2e-3 * np.abs(g64)).mean(), err.sum() / np.abs(g64).sum()


@pytest.mark.parametrize("seed,B,H,W", [(505, 1, 192, 640), (404, 2, 64, 96)])
def test_loss_path_error_against_float64(FD, seed, B, H, W):
    """North-star tolerance, stated against ground truth: the float64 run of the oracle.  The reference's own float32 arithmetic
    (= the float32 oracle) is itself 2e-4 ... 1e-3 (relative L1) away from the float64 gradient maps, because a pixel whose argmin
    / clamp / |.| branch sits within rounding of a tie takes either branch; the HIP kernels (both the per-scale ones that mimic the
    reference's operation order and the multi-scale one that does not) must be no further from float64 than the float32 reference
    is (x1.5 + a floor of 3e-4 relative L1 / 0.2 % of the entries: which handful of near-tie pixels flips differs between any two
    evaluation orders), and their scalar losses within 1e-6."""
    opt = OT.default_opt(height=H, width=W)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp0 = gin.disp_pyramid(rng, B, H, W)
    poses = {f: gin.small_poses(rng, B) for f in (-1, 1)}
    T0 = {f: OL.transformation_from_parameters(*poses[f], invert=(f < 0)) for f in (-1, 1)}
    noise = [torch.from_numpy(np.random.RandomState(1000 + seed + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]

    def oracle(dt):
        i2 = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
        d = {s: disp0[("disp", s)].to(dt).clone().requires_grad_(True) for s in range(4)}
        Tq = {f: T0[f].to(dt).clone().requires_grad_(True) for f in T0}
        terms, _ = _oracle_photo_terms(opt, i2, d, Tq, [n.to(dt) for n in noise])
        g = torch.autograd.grad(sum(terms[s][0] for s in range(4)), [d[s] for s in range(4)])
        return [float(terms[s][0]) for s in range(4)], [x.double().numpy() for x in g]

    def hip(ms):
        d = {s: dev(disp0[("disp", s)]).requires_grad_(True) for s in range(4)}
        Tq = {f: dev(T0[f]).requires_grad_(True) for f in T0}
        if ms:
            photo = _hip_photo_terms_ms(FD, opt, inp, d, Tq, noise)[0]
        else:
            photo = [r[0] for r in _hip_photo_terms(FD, opt, inp, d, Tq, noise, materialize=False)]
        g = torch.autograd.grad(sum(photo), [d[s] for s in range(4)])
        return [float(p) for p in photo], [cpu(x) for x in g]

    v64, g64 = oracle(torch.float64)
    v32, g32 = oracle(torch.float32)
    for name, (v, g) in (("per-scale", hip(False)), ("multi-scale", hip(True))):
        for s in range(4):
            assert abs(v[s] - v64[s]) <= 1e-6 * abs(v64[s]), "%s loss s%d: %.9g vs float64 %.9g" % (name, s, v[s], v64[s])
            bad_ref, l1_ref = _grad_error_stats(g32[s], g64[s])
            bad, l1 = _grad_error_stats(g[s], g64[s])
            print("[vs float64] %-11s s%d: %.3f%% of gradient entries out of tolerance (float32 reference %.3f%%), rel-L1 %.2e "
                  "(float32 reference %.2e)" % (name, s, 100 * bad, 100 * bad_ref, l1, l1_ref))
            assert bad <= 1.5 * bad_ref + 2e-3, "%s s%d: %.3f%% vs %.3f%% for the float32 reference" % (name, s, 100 * bad, 100 * bad_ref)
            assert l1 <= 1.5 * l1_ref + 3e-4, "%s s%d: rel-L1 %.3g vs %.3g for the float32 reference" % (name, s, l1, l1_ref)

def _golden_loss_case(FD, g, seed, B, H, W, kernel, empty_si=None):
    """Inputs of tests/golden/make_golden.py::gold_losses -> (per-scale loss terms, total, leaves) from one of the two HIP
    implementations ("ms": fd_photo_ms_* = what the trainer and the bench run; "per_scale": fd_photo_fwd_ex / bwd_ex)."""
    opt = OT.default_opt(height=H, width=W)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp0 = gin.disp_pyramid(rng, B, H, W)
    gin.make_si_mask_empty(inp, disp0, empty_si)
    d_g = {s: dev(disp0[("disp", s)]).requires_grad_(True) for s in range(4)}
    T_g = {f: torch.from_numpy(g["T%d" % f]).cuda().requires_grad_(True) for f in (-1, 1)}
    torch.manual_seed(int(g["noise_seed"]))
    noise = [torch.randn(B, 2, H, W) for _ in range(4)]
    np.testing.assert_array_equal(noise[0].numpy().reshape(-1)[:16], g["noise_head"])
    if kernel == "ms":
        photo, si, sel = _hip_photo_terms_ms(FD, opt, inp, d_g, T_g, noise)
    else:
        res = _hip_photo_terms(FD, opt, inp, d_g, T_g, noise, materialize=True)
        photo, si, sel = [r[0] for r in res], [r[1] for r in res], [r[2] for r in res]
        for s in range(4):           # the materialised by-products against the reference's ("depth", 0, s) / ("color", f, s)
            depth, color = res[s][3], res[s][5]
            if "depth%d" % s in g:
                assert_close(cpu(depth), g["depth%d" % s], rtol=1e-5, atol=1e-6, what="depth s%d" % s)
                for i, f in enumerate((-1, 1)):
                    assert_close(cpu(color[i]), g["color%d_%d" % (f, s)], rtol=1e-4, atol=2e-5, what="color f%d s%d" % (f, s))
            elif "depth%d_sub" % s in g:
                assert_close(cpu(depth)[:, :, ::16, ::16], g["depth%d_sub" % s], rtol=1e-5, atol=1e-6, what="depth s%d" % s)
                for i, f in enumerate((-1, 1)):
                    assert_close(cpu(color[i])[:, :, ::16, ::16], g["color%d_%d_sub" % (f, s)], rtol=1e-4, atol=2e-5,
                                 what="color f%d s%d" % (f, s))
    loss_s = [photo[s] + opt.disparity_smoothness * FD.normalized_smooth_loss(d_g[s], dev(inp[("color", 0, s)])) / (2 ** s)
              for s in range(4)]
    total = sum(loss_s[s] + si[s] for s in range(4)) / 4
    return loss_s, si, sel, total, [d_g[s] for s in range(4)] + [T_g[-1], T_g[1]]


@pytest.mark.parametrize("name,seed,B,H,W", [("losses_b2_64x96", 404, 2, 64, 96), ("losses_b1_192x640", 505, 1, 192, 640)])
@pytest.mark.parametrize("kernel", ["ms", "per_scale"])
def test_default_loss_kernels_vs_reference_golden(FD, golden, name, seed, B, H, W, kernel):
    """BOTH HIP implementations of generate_images_pred + compute_losses (trainer.py:425-596) - the all-scales kernel the trainer
    and bench.py run by default, and the per-scale kernels of the flag variants - directly against what the REFERENCE produced on
    these inputs (tests/golden/make_golden.py::gold_losses), at 64x96 and at the full 192x640."""
    g = golden(name)
    loss_s, si, sel, total, leaves = _golden_loss_case(FD, g, seed, B, H, W, kernel)
    flips = 0
    for s in range(4):
        assert_close(cpu(loss_s[s]), g["L/loss_%d" % s], rtol=1e-4, atol=1e-7, what="loss/%d" % s)
        assert_close(cpu(si[s]), g["L/loss_si_loss%d" % s], rtol=1e-4, atol=1e-7, what="si_loss%d" % s)
        diff = (cpu(sel[s]) > 1).astype(np.uint8) != g["idsel%d" % s]
        flips += int(diff.sum())
        assert diff.mean() <= 3e-4
    assert_close(cpu(total), g["L/loss"], rtol=1e-4, atol=1e-7, what="total loss")
    got = grads(total, leaves)
    for s in range(4):
        sc = np.abs(g["g_disp%d" % s]).max()
        assert_mostly_close(got[s], g["g_disp%d" % s], rtol=2e-3, atol=2e-4 * sc, what="g disp%d vs reference (%s)" % (s, kernel),
                            max_bad_frac=max(1e-2, 4.0 / got[s].size))
        flips += int((np.abs(got[s] - g["g_disp%d" % s]) > 1e-3 * sc).sum())
    # The pose gradients are sums over every pixel.  Without a single argmin / clamp / |.| branch flip they agree to ~1e-6 of their
    # largest entry; each pixel whose branch sits within rounding of a tie and flips (counted above: identity-selection mismatches
    # + isolated O(1) differences in the disparity gradient, a few of the 0.5 M pixel visits at 192x640) moves them by up to a percent
    # of it - the same allowance as in test_multiscale_photo_loss_vs_oracle.
    for i, f in ((4, -1), (5, 1)):
        sc = np.abs(g["g_T%d" % f]).max()
        assert_close(got[i], g["g_T%d" % f], rtol=1e-3, atol=(3e-2 if flips else 5e-4) * sc, what="g T%d vs reference (%d flips)" % (f, flips))


@pytest.mark.parametrize("mode", ["all", "scale2"])
@pytest.mark.parametrize("kernel", ["ms", "per_scale"])
def test_empty_lidar_mask_vs_reference_golden(FD, golden, mode, kernel):
    """trainer.py:577-589 with no LiDAR return inside the validity mask (no returns at all / none at the 1/4 scale): the reference
    takes the mean of an empty selection - si_loss of that scale and the total are NaN, all gradients stay finite (golden:
    make_golden.py gold_losses(empty_si=...)).  Same NaN pattern, same values elsewhere, same gradients, for both kernels - this is
    the state a from-scratch run on synthetic frames drifts into (bench.py)."""
    g = golden("losses_emptysi_%s_b2_64x96" % mode)
    loss_s, si, sel, total, leaves = _golden_loss_case(FD, g, 404, 2, 64, 96, kernel, empty_si=mode)
    want_nan = [0, 1, 2, 3] if mode == "all" else [2]
    for s in range(4):
        assert bool(torch.isnan(si[s])) == (s in want_nan), "si_loss%d: %r" % (s, float(si[s]))
        assert_close(cpu(si[s]), g["L/loss_si_loss%d" % s], rtol=1e-4, atol=1e-7, what="si_loss%d" % s)     # NaN == NaN here
        assert_close(cpu(loss_s[s]), g["L/loss_%d" % s], rtol=1e-4, atol=1e-7, what="loss/%d" % s)
    assert bool(torch.isnan(total)) and np.isnan(g["L/loss"])
    got = grads(total, leaves)
    for s in range(4):
        assert np.isfinite(got[s]).all(), "d loss / d disp%d has non-finite entries" % s
        sc = np.abs(g["g_disp%d" % s]).max()
        # 8x12 maps at scale 3: two entries within rounding of a clamp / |.| tie are already 1 % of the map
        assert_mostly_close(got[s], g["g_disp%d" % s], rtol=2e-3, atol=2e-4 * sc, what="g disp%d vs reference (%s, %s)" % (s, kernel, mode),
                            max_bad_frac=max(1e-2, 4.0 / got[s].size))
    for i, f in ((4, -1), (5, 1)):
        assert np.isfinite(got[i]).all()
        assert_close(got[i], g["g_T%d" % f], rtol=1e-3, atol=5e-4 * np.abs(g["g_T%d" % f]).max(), what="g T%d vs reference" % f)


@pytest.mark.parametrize("kernel", ["ms", "per_scale"])
def test_empty_lidar_mask_in_one_stacked_micro_batch(FD, kernel):
    """Two micro-batches stacked into one pass (groups = 2, Trainer.train_step), the first without any LiDAR return: the reference
    runs them one after the other (trainer.py:237-248) - micro-batch 0 gives a NaN si_loss and finite gradients, micro-batch 1 is
    unaffected - so the stacked value (mean over the micro-batches) is NaN at every scale while micro-batch 1's disparities still
    receive exactly half of their stand-alone SI gradient.  Oracle: the two micro-batches evaluated separately."""
    B, H, W, seed = 2, 64, 96, 404
    opt = OT.default_opt(height=H, width=W)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp0 = gin.disp_pyramid(rng, B, H, W)
    gin.make_si_mask_empty(inp, disp0, "sample0")
    poses = {f: gin.small_poses(rng, B) for f in (-1, 1)}
    T0 = {f: OL.transformation_from_parameters(*poses[f], invert=(f < 0)) for f in (-1, 1)}
    noise = [torch.from_numpy(np.random.RandomState(1000 + seed + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
    # oracle: micro-batch b on its own, total = sum_b (sum_s photo + si) / 2
    want_g, want_si = [], []
    for b in range(B):
        sl = slice(b, b + 1)
        i_b = {k: (v[sl] if torch.is_tensor(v) else v) for k, v in inp.items()}
        d_b = {s: disp0[("disp", s)][sl].clone().requires_grad_(True) for s in range(4)}
        T_b = {f: T0[f][sl].clone().requires_grad_(True) for f in T0}
        terms, _ = _oracle_photo_terms(opt, i_b, d_b, T_b, [n[sl] for n in noise])
        tot = sum(terms[s][0] + terms[s][1] for s in range(4)) / B
        want_g.append(grads(tot, [d_b[s] for s in range(4)]))
        want_si.append([float(terms[s][1]) for s in range(4)])
    assert all(np.isnan(v) for v in want_si[0]) and all(np.isfinite(v) for v in want_si[1])
    d_g = {s: dev(disp0[("disp", s)]).requires_grad_(True) for s in range(4)}
    T_g = {f: dev(T0[f]).requires_grad_(True) for f in T0}
    if kernel == "ms":
        photo, si, _ = _hip_photo_terms_ms(FD, opt, inp, d_g, T_g, noise, groups=2)
    else:
        res = _hip_photo_terms(FD, opt, inp, d_g, T_g, noise, materialize=False, groups=2)
        photo, si = [r[0] for r in res], [r[1] for r in res]
    assert all(bool(torch.isnan(v)) for v in si), [float(v) for v in si]
    got = grads(sum(photo[s] + si[s] for s in range(4)), [d_g[s] for s in range(4)])
    for s in range(4):
        assert np.isfinite(got[s]).all()
        want = np.concatenate([want_g[0][s], want_g[1][s]], 0)
        assert_mostly_close(got[s], want, rtol=2e-3, atol=2e-4 * np.abs(want).max(), what="d loss / d disp s%d (%s)" % (s, kernel),
                            max_bad_frac=max(1e-2, 4.0 / got[s].size))


def test_multiscale_photo_loss_matches_per_scale_kernels(FD):
    """The two HIP implementations agree with each other (value, selection, gradients) incl. a LiDAR term on scale 0 only."""
    B, H, W, seed = 2, 64, 96, 404
    opt = OT.default_opt(height=H, width=W)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp0 = gin.disp_pyramid(rng, B, H, W)
    poses = {f: gin.small_poses(rng, B) for f in (-1, 1)}
    T0 = {f: OL.transformation_from_parameters(*poses[f], invert=(f < 0)) for f in (-1, 1)}
    noise = [torch.from_numpy(np.random.RandomState(1000 + seed + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
    da = {s: dev(disp0[("disp", s)]).requires_grad_(True) for s in range(4)}
    db = {s: dev(disp0[("disp", s)]).requires_grad_(True) for s in range(4)}
    Ta = {f: dev(T0[f]).requires_grad_(True) for f in T0}
    Tb = {f: dev(T0[f]).requires_grad_(True) for f in T0}
    res = _hip_photo_terms(FD, opt, inp, da, Ta, noise, materialize=False)
    photo, si, sel = _hip_photo_terms_ms(FD, opt, inp, db, Tb, noise, beam_scales=(0,))
    assert si[1] is None and si[2] is None and si[3] is None
    tot_a = sum((1 + s) * res[s][0] for s in range(4)) + 0.7 * res[0][1]
    tot_b = sum((1 + s) * photo[s] for s in range(4)) + 0.7 * si[0]
    for s in range(4):
        assert (cpu(res[s][2]) != cpu(sel[s])).mean() <= 3e-4
        assert_close(cpu(photo[s]), cpu(res[s][0]), rtol=2e-5, atol=1e-7, what="photo s%d" % s)
    assert_close(cpu(si[0]), cpu(res[0][1]), rtol=2e-5, atol=1e-7, what="si s0")
    ga = grads(tot_a, [da[s] for s in range(4)] + [Ta[-1], Ta[1]])
    gb = grads(tot_b, [db[s] for s in range(4)] + [Tb[-1], Tb[1]])
    for s in range(4):
        # 8x12 maps at scale 3: one pixel whose clamp / sign branch differs between the two arithmetic orders is already 0.5 %
        assert_mostly_close(gb[s], ga[s], rtol=2e-3, atol=2e-4 * np.abs(ga[s]).max(), what="d disp s%d" % s, max_bad_frac=3e-2)
    for i in (4, 5):
        assert_close(gb[i], ga[i], rtol=1e-3, atol=3e-2 * np.abs(ga[i]).max(), what="d T")


def test_photo_loss_identity_pose_property(FD):
    """Size-independent property at full size: with T = I the sampling grid is the pixel grid normalised
    by (W-1,H-1) (layers.py:224-226) and the depth cancels out of the projection; because grid_sample
    runs with align_corners=False the sampled position is x*W/(W-1)-0.5 (the reference's own quirk),
    i.e. a sub-pixel stretch, so the loss is small but not zero."""
    B, H, W = 2, 192, 640
    inp, rng = gin.batch_inputs(1, B, H, W)
    disp = dev(gin.disp_pyramid(rng, B, H, W)[("disp", 0)])
    I = torch.eye(4).repeat(B, 1, 1).cuda()
    tgt = dev(inp[("color", 0, 0)])
    photo, si, sel, depth, sample, color = FD.photo_loss(disp, [I, I], dev(inp[("K", 0)]), dev(inp[("inv_K", 0)]),
                                                         [tgt, tgt], tgt, None, None, None, FD.PhotoOptions(), True)
    assert 0 <= float(photo) < 0.1
    want_c = F.grid_sample(inp[("color", 0, 0)], cpu_grid(H, W, B), padding_mode="border", align_corners=False)
    assert_close(cpu(color[0]), want_c.numpy(), rtol=0, atol=1e-4, what="identity-pose warp")
    assert_close(cpu(color[1]), cpu(color[0]), rtol=0, atol=0, what="both frames identical")
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    want = torch.stack([(xs / (W - 1) - 0.5) * 2, (ys / (H - 1) - 0.5) * 2], -1)
    assert_close(cpu(sample[0][0]), want.numpy(), rtol=0, atol=1e-4, what="identity sampling grid")


def cpu_grid(H, W, B):
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    return torch.stack([(xs / (W - 1) - 0.5) * 2, (ys / (H - 1) - 0.5) * 2], -1)[None].repeat(B, 1, 1, 1)


def test_ops_refuse_cpu_tensors(FD):
    with pytest.raises(RuntimeError):
        FD.ssim(torch.rand(1, 3, 8, 8), torch.rand(1, 3, 8, 8))


def test_integration_stub_functions(FD, monkeypatch):
    """The ctypes stub of INTEGRATION.md section 2 (what a reference maintainer would paste), executed as written: its ``ssim``,
    ``get_4beam_2channel`` and ``get_4beam`` against the package's own wrappers (themselves checked against the oracle above)."""
    import os
    import test_abi
    from fusiondepth_amd import functional as F_
    monkeypatch.chdir(test_abi.ROOT)
    ns = {}
    exec(compile(test_abi.integration_stub(), "INTEGRATION.md", "exec"), ns)
    rng = np.random.RandomState(5)
    x, y = (torch.from_numpy(rng.rand(2, 3, 32, 64).astype(np.float32)).cuda() for _ in range(2))
    assert torch.equal(ns["ssim"](x, y), F_.ssim(x, y))
    beam = torch.zeros(2, 1, 192, 640)
    beam[:, 0, 100:180:20, 4:636:3] = torch.from_numpy(rng.uniform(0.05, 0.65, (2, 4, 211)).astype(np.float32))
    beam = beam.cuda()
    assert torch.equal(ns["get_4beam_2channel"](beam), F_.scatter_2channel(beam))
    velo, P = gin.lidar_scan(21, n_points=5000)
    pts = torch.from_numpy(velo).cuda()
    got = ns["get_4beam"](pts, torch.from_numpy(np.ascontiguousarray(P, dtype=np.float64)).cuda(), 375, 1242)
    want = F_.velo_rasterize(pts, P, 375, 1242, (384, 1280))
    torch.cuda.synchronize()
    assert torch.equal(got.reshape(-1), want.reshape(-1)) and float(got.abs().sum()) > 0
