"""Pin the CPU oracle against golden vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_close, check_grad_compact
from oracle import layers as OL
from oracle import networks as ON
from oracle import scatter as OS
from oracle import trainer as OT

# The oracle uses the same torch CPU primitives as the reference => expect (near) bit equality.
TIGHT = dict(rtol=1e-6, atol=1e-7)


def npy(t):
    return t.detach().numpy()


def test_layers_against_reference(golden):
    g = golden("layers_b2_32x64")
    B, H, W = 2, 32, 64
    inp, rng = gin.batch_inputs(101, B, H, W)
    disp = gin.disp_pyramid(rng, B, H, W)[("disp", 0)].requires_grad_(True)
    sd, depth = OL.disp_to_depth(disp, 0.1, 100.0)
    assert_close(npy(sd), g["d2d_scaled"], what="scaled_disp", **TIGHT)
    assert_close(npy(depth), g["d2d_depth"], what="depth", **TIGHT)

    aa, tr = gin.small_poses(rng, B)
    aa.requires_grad_(True), tr.requires_grad_(True)
    cot_T = torch.from_numpy(rng.randn(B, 4, 4).astype(np.float32))
    assert_close(npy(cot_T), g["T_cot"], **TIGHT)
    for inv in (False, True):
        tag = "inv" if inv else "fwd"
        M = OL.transformation_from_parameters(aa, tr, invert=inv)
        ga, gt = torch.autograd.grad((M * cot_T).sum(), [aa, tr])
        assert_close(npy(M), g["T_" + tag], what="T_" + tag, **TIGHT)
        assert_close(npy(ga), g["T_%s_gaa" % tag], rtol=1e-5, atol=1e-6, what="gaa")
        assert_close(npy(gt), g["T_%s_gtr" % tag], rtol=1e-5, atol=1e-6, what="gtr")

    K, inv_K = inp[("K", 0)], inp[("inv_K", 0)]
    T = torch.from_numpy(g["pj_T"]).requires_grad_(True)
    depth_in = depth.detach().clone().requires_grad_(True)
    pts = OL.backproject_depth(depth_in, inv_K)
    grid = OL.project_3d(pts, K, T, H, W)
    cot_g = torch.from_numpy(rng.randn(B, H, W, 2).astype(np.float32))
    gd, gT = torch.autograd.grad((grid * cot_g).sum(), [depth_in, T])
    assert_close(npy(pts), g["bp_points"], what="backproject", **TIGHT)
    assert_close(npy(grid), g["pj_grid"], what="project grid", rtol=1e-6, atol=1e-6)
    assert_close(npy(gd), g["pj_gdepth"], rtol=1e-5, atol=1e-6, what="g depth")
    assert_close(npy(gT), g["pj_gT"], rtol=1e-4, atol=1e-4, what="g T")

    x = inp[("color", 0, 0)].clone().requires_grad_(True)
    y = inp[("color", 1, 0)].clone().requires_grad_(True)
    s = OL.ssim(x, y)
    cot_s = torch.from_numpy(rng.rand(B, 3, H, W).astype(np.float32))
    gx, gy = torch.autograd.grad((s * cot_s).sum(), [x, y])
    assert_close(npy(s), g["ssim"], what="ssim", **TIGHT)
    assert_close(npy(gx), g["ssim_gx"], rtol=1e-5, atol=1e-6, what="ssim gx")
    assert_close(npy(gy), g["ssim_gy"], rtol=1e-5, atol=1e-6, what="ssim gy")

    d = disp.detach().clone().requires_grad_(True)
    sm = OL.get_smooth_loss(d, inp[("color", 0, 0)])
    assert_close(npy(sm), g["smooth"], what="smooth", **TIGHT)
    assert_close(npy(torch.autograd.grad(sm, d)[0]), g["smooth_gdisp"], rtol=1e-5, atol=1e-8, what="smooth grad")

    xin = torch.from_numpy(g["cb_x"]).requires_grad_(True)
    w = torch.from_numpy(g["cb_w"]).requires_grad_(True)
    b = torch.from_numpy(g["cb_b"]).requires_grad_(True)
    yo = OL.conv_block(xin, w, b)
    gxx, gw, gb = torch.autograd.grad((yo * torch.from_numpy(g["cb_cot"])).sum(), [xin, w, b])
    assert_close(npy(yo), g["cb_y"], rtol=1e-5, atol=1e-6, what="ConvBlock")
    assert_close(npy(gxx), g["cb_gx"], rtol=1e-5, atol=1e-5, what="ConvBlock gx")
    assert_close(npy(gw), g["cb_gw"], rtol=1e-4, atol=1e-4, what="ConvBlock gw")
    assert_close(npy(gb), g["cb_gb"], rtol=1e-4, atol=1e-4, what="ConvBlock gb")
    y3 = OL.conv3x3(xin, torch.from_numpy(g["c3_w"]), torch.from_numpy(g["c3_b"]), use_refl=False)
    assert_close(npy(y3), g["c3_y"], rtol=1e-5, atol=1e-6, what="Conv3x3 zero pad")
    assert_close(npy(OL.upsample(xin)), g["up_y"], what="upsample", **TIGHT)
    assert_close(npy(OL.cat_xy(depth.detach(), inv_K)), g["catxy"], rtol=1e-6, atol=1e-6, what="Cat_xy")

    errs = OL.compute_depth_errors(torch.from_numpy(g["errs_gt"]), torch.from_numpy(g["errs_pred"]))
    assert_close([float(v) for v in errs], g["errs"], rtol=1e-6, atol=0, what="depth errors")


def _decoder_setup():
    B, H, W = 2, 64, 96
    rng = np.random.RandomState(202)
    ch = np.array([64, 64, 128, 256, 512])
    feats, beams = gin.feature_pyramids(rng, B, H, W, ch)
    return B, H, W, rng, ch, feats, beams


def test_depth_and_pose_decoders_against_reference(golden):
    g = golden("decoders_b2_64x96")
    B, H, W, rng, ch, feats, beams = _decoder_setup()
    for t in feats + beams:
        t.requires_grad_(True)
    dec = gin.fill_params(ON.DepthDecoder(ch), 11)
    o = dec(feats, beam_features=beams)
    cots = {s: torch.from_numpy(rng.randn(*o[("disp", s)].shape).astype(np.float32)) for s in range(4)}
    for s in range(4):
        assert_close(npy(cots[s]), g["dec_cot%d" % s], **TIGHT)
        assert_close(npy(o[("disp", s)]), g["dec_disp%d" % s], rtol=1e-5, atol=1e-6, what="disp%d" % s)
    loss = sum((o[("disp", s)] * cots[s]).sum() for s in range(4))
    grads = torch.autograd.grad(loss, feats + beams + list(dec.parameters()))
    for i in range(5):
        check_grad_compact(g, "dec_gfeat%d" % i, npy(grads[i]), rtol=1e-4, atol=1e-5)
        check_grad_compact(g, "dec_gbeam%d" % i, npy(grads[5 + i]), rtol=1e-4, atol=1e-5)
    for (k, _), gv in zip(dec.named_parameters(), grads[10:]):
        check_grad_compact(g, "dec_g/" + k.replace(".", "/"), npy(gv), rtol=1e-4, atol=1e-4)
    o2 = dec([f.detach() for f in feats])
    assert_close(npy(o2[("disp", 0)]), g["dec_nobeam_disp0"], rtol=1e-5, atol=1e-6, what="no-beam disp0")

    pose = gin.fill_params(ON.PoseDecoder(ch, num_input_features=1, num_frames_to_predict_for=2), 12)
    f4 = feats[4].detach().clone().requires_grad_(True)
    b4 = beams[4].detach().clone().requires_grad_(True)
    aa, tr = pose([[None] * 4 + [f4]], beam_inputs=[[None] * 4 + [b4]])
    assert_close(npy(aa), g["pose_aa"], rtol=1e-5, atol=1e-8, what="axisangle")
    assert_close(npy(tr), g["pose_tr"], rtol=1e-5, atol=1e-8, what="translation")
    gr = torch.autograd.grad((aa * torch.from_numpy(g["pose_cot_a"])).sum() + (tr * torch.from_numpy(g["pose_cot_t"])).sum(),
                             [f4, b4] + list(pose.parameters()))
    assert_close(npy(gr[0]), g["pose_gf4"], rtol=1e-4, atol=1e-7, what="pose g f4")
    for (k, _), gv in zip(pose.named_parameters(), gr[2:]):
        check_grad_compact(g, "pose_g/" + k.replace(".", "/"), npy(gv), rtol=1e-4, atol=1e-6)

    pcnn = gin.fill_params(ON.PoseCNN(2), 13)
    xin = torch.from_numpy(np.random.RandomState(213).rand(B, 6, H, W).astype(np.float32))
    a2, t2 = pcnn(xin)
    assert_close(npy(a2), g["posecnn_aa"], rtol=1e-5, atol=1e-8, what="posecnn aa")
    assert_close(npy(t2), g["posecnn_tr"], rtol=1e-5, atol=1e-8, what="posecnn tr")


def test_refine_decoder_variant_against_reference(golden):
    g = golden("decoder_refine_b2_64x96")
    B, H, W, rng, ch, feats, beams = _decoder_setup()
    dec2 = gin.fill_params(ON.DepthDecoder(ch, road=True, catxy=True, deep=True), 14)
    assert sum(p.numel() for p in dec2.parameters()) == int(g["num_params"]) == 9547052
    rng2 = np.random.RandomState(214)
    dm = {("disp", s): torch.from_numpy(rng2.rand(B, 6, H // 2 ** s, W // 2 ** s).astype(np.float32)) for s in range(4)}
    o = dec2(feats, beam_features=beams, depth_maps=dm, tanh=True)
    for s in range(4):
        assert_close(npy(o[("disp", s)]), g["disp%d" % s], rtol=1e-5, atol=1e-6, what="refine disp%d" % s)


def _loss_case(g, seed, B, H, W, empty_si=None, **opt_over):
    opt = OT.default_opt(height=H, width=W, **opt_over)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp = gin.disp_pyramid(rng, B, H, W)
    gin.make_si_mask_empty(inp, disp, empty_si)
    outputs, leaves = {}, []
    for s in range(4):
        outputs[("disp", s)] = disp[("disp", s)].clone().requires_grad_(True)
        leaves.append(outputs[("disp", s)])
    for f in (-1, 1):
        aa, tr = gin.small_poses(rng, B)
        T = OL.transformation_from_parameters(aa, tr, invert=(f < 0)).detach().requires_grad_(True)
        assert_close(npy(T), g["T%d" % f], what="T", **TIGHT)
        outputs[("cam_T_cam", 0, f)] = T
        leaves.append(T)
    torch.manual_seed(int(g["noise_seed"]))
    noise = [torch.randn(B, 2, H, W) for _ in range(4)]
    assert_close(npy(noise[0]).reshape(-1)[:16], g["noise_head"], what="noise stream", **TIGHT)
    OT.generate_images_pred(opt, inp, outputs)
    losses = OT.compute_losses(opt, inp, outputs, noise)
    grads = torch.autograd.grad(losses["loss"], leaves)
    return opt, outputs, losses, grads


@pytest.mark.parametrize("name,seed,B,H,W", [("losses_b2_64x96", 404, 2, 64, 96), ("losses_b1_192x640", 505, 1, 192, 640)])
def test_loss_path_against_reference(golden, name, seed, B, H, W):
    g = golden(name)
    opt, outputs, losses, grads = _loss_case(g, seed, B, H, W)
    for k, v in losses.items():
        assert_close(float(v), g["L/" + k.replace("/", "_")], rtol=1e-6, atol=1e-8, what=k)
    for s in range(4):
        assert_close(npy(grads[s]), g["g_disp%d" % s], rtol=1e-5, atol=1e-9, what="g disp%d" % s)
        assert (npy(outputs["identity_selection/%d" % s]).astype(np.uint8) == g["idsel%d" % s]).all()
        if "depth%d" % s in g:
            assert_close(npy(outputs[("depth", 0, s)]), g["depth%d" % s], what="depth", **TIGHT)
            for f in (-1, 1):
                assert_close(npy(outputs[("sample", f, s)]), g["sample%d_%d" % (f, s)], rtol=1e-6, atol=1e-6, what="sample")
                assert_close(npy(outputs[("color", f, s)]), g["color%d_%d" % (f, s)], rtol=1e-5, atol=1e-6, what="color")
        else:
            assert_close(npy(outputs[("depth", 0, s)])[:, :, ::16, ::16], g["depth%d_sub" % s], what="depth", **TIGHT)
    assert_close(npy(grads[4]), g["g_T-1"], rtol=1e-4, atol=1e-6, what="g T-1")
    assert_close(npy(grads[5]), g["g_T1"], rtol=1e-4, atol=1e-6, what="g T+1")


def test_loss_path_flags_no_ssim_no_automask(golden):
    g = golden("losses_nossim_noautomask_b2_64x96")
    opt, outputs, losses, grads = _loss_case(g, 606, 2, 64, 96, no_ssim=True, disable_automasking=True)
    for k, v in losses.items():
        assert_close(float(v), g["L/" + k.replace("/", "_")], rtol=1e-6, atol=1e-8, what=k)
    for s in range(4):
        assert_close(npy(grads[s]), g["g_disp%d" % s], rtol=1e-5, atol=1e-9, what="g disp%d" % s)


@pytest.mark.parametrize("mode", ["all", "scale2"])
def test_loss_path_with_empty_lidar_mask(golden, mode):
    """trainer.py:577-589 when no LiDAR return passes the validity mask (everywhere / at one scale): the reference's si_loss of
    that scale and the total are NaN, every gradient stays finite (an empty selection passes nothing back).  The oracle must show
    the same NaN pattern and the same gradients."""
    g = golden("losses_emptysi_%s_b2_64x96" % mode)
    opt, outputs, losses, grads = _loss_case(g, 404, 2, 64, 96, empty_si=mode)
    nan_keys = [k for k in losses if np.isnan(float(losses[k]))]
    assert nan_keys == (["loss/si_loss%d" % s for s in range(4)] + ["loss"] if mode == "all" else ["loss/si_loss2", "loss"])
    for k, v in losses.items():
        assert_close(float(v), g["L/" + k.replace("/", "_")], rtol=1e-6, atol=1e-8, what=k)       # checks the NaN pattern too
    for s in range(4):
        assert np.isfinite(npy(grads[s])).all()
        assert_close(npy(grads[s]), g["g_disp%d" % s], rtol=1e-5, atol=1e-9, what="g disp%d" % s)
    assert_close(npy(grads[4]), g["g_T-1"], rtol=1e-4, atol=1e-6, what="g T-1")
    assert_close(npy(grads[5]), g["g_T1"], rtol=1e-4, atol=1e-6, what="g T+1")


def test_scatter_c_oracle_bit_exact_vs_reference(golden):
    g = golden("scatter_192x640")
    for i in range(3):
        d, c = OS.scatter_2channel_c(g["beam%d" % i])
        assert np.array_equal(c, g["conf%d" % i]), "confidence map %d" % i
        assert np.array_equal(d, g["depth%d" % i]), "expanded depth %d: max diff %g" % (i, np.abs(d - g["depth%d" % i]).max())
        d2, c2 = OS.scatter_2channel_np(g["beam%d" % i])
        assert np.array_equal(c2, c)
        assert np.array_equal(d2, d), "numpy formulation differs from sequential C by %g" % np.abs(d2 - d).max()


def test_scatter_edge_cases():
    empty = np.zeros((192, 640), np.float32)
    d, c = OS.scatter_2channel_c(empty)
    assert not d.any() and not c.any()
    outside = empty.copy()
    outside[10, 10] = 0.5          # outside the ROI: ignored (gen2channel.py:64-65)
    outside[190, 300] = 0.5
    outside[100, 1] = 0.5
    d, c = OS.scatter_2channel_c(outside)
    assert not d.any() and not c.any()
    one = empty.copy()
    one[100, 100] = 0.25
    d, c = OS.scatter_2channel_c(one)
    assert c[100, 100] == 1 and c[99, 100] == 0.5 and c[101, 100] == 0.5
    assert c[100, 99] == 0 and c[100, 101] == 0            # never purely horizontal
    third = np.float32(1.0 / 3.0)
    for rr, cc in [(98, 100), (102, 100), (99, 99), (99, 101), (101, 99), (101, 101)]:
        assert c[rr, cc] == third and d[rr, cc] == np.float32(0.25)
    assert (c > 0).sum() == 9


def test_resnet_trunk_structure():
    """'parity unpinned' piece: structural pins only — key names, shapes, param counts."""
    enc = ON.ResnetEncoder(18, False)
    keys = list(enc.state_dict().keys())
    assert keys[0] == "encoder.conv1.weight" and "encoder.layer2.0.downsample.1.running_var" in keys
    assert "encoder.fc.weight" in keys
    n = lambda m: sum(p.numel() for k, p in m.named_parameters() if ".fc." not in k)
    assert n(ON.ResnetEncoder(18, False)) == 11176512
    assert n(ON.ResnetEncoder(18, False, beam_encoder=True)) == 11173376
    assert n(ON.ResnetEncoder(18, False, num_input_images=2)) == 11185920
    assert n(ON.ResnetEncoder(18, False, num_input_images=2, beam_encoder=True)) == 11179648
    dec = ON.DepthDecoder(enc.num_ch_enc)
    assert sum(p.numel() for p in dec.parameters()) == 3152724
    pose = ON.PoseDecoder(enc.num_ch_enc, 1, 2)
    assert sum(p.numel() for p in pose.parameters()) == 1314572
    e50 = ON.ResnetEncoder(50, False)
    assert list(e50.num_ch_enc) == [64, 256, 512, 1024, 2048]
    feats = enc(torch.rand(1, 3, 64, 96))
    assert [tuple(f.shape[1:]) for f in feats] == [(64, 32, 48), (64, 16, 24), (128, 8, 12), (256, 4, 6), (512, 2, 3)]


@pytest.mark.parametrize("num_layers", [18, 50])
def test_resnet_trunk_matches_an_independent_resnet(num_layers):
    """torchvision (the reference's `models.resnet18/50`, networks/resnet_encoder.py:61-75) is not installed here, so the
    oracle's ResNetTrunk is pinned against another public implementation of the same architecture that IS: Hugging Face
    `transformers.ResNetModel` (basic / bottleneck layers, stride on the 3x3 = ResNet v1.5 like torchvision).  Same weights ->
    the same five feature maps, in training mode (batch statistics) and in eval mode, values and input gradients."""
    tf = pytest.importorskip("transformers")
    torch.manual_seed(3)
    enc = ON.ResnetEncoder(num_layers, False)
    gin.fill_params(enc, 41)
    trunk = enc.encoder
    wide = num_layers > 34
    cfg = tf.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048] if wide else [64, 128, 256, 512],
                          depths={18: [2, 2, 2, 2], 50: [3, 4, 6, 3]}[num_layers], layer_type="bottleneck" if wide else "basic",
                          hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    hf = tf.ResNetModel(cfg)
    src, sd = trunk.state_dict(), {}

    def conv_bn(dst, conv, bn):
        sd[dst + ".convolution.weight"] = src[conv + ".weight"]
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            sd[dst + ".normalization." + k] = src[bn + "." + k]

    conv_bn("embedder.embedder", "conv1", "bn1")
    for si in range(4):
        for bj, block in enumerate(getattr(trunk, "layer%d" % (si + 1))):
            o, h = "layer%d.%d" % (si + 1, bj), "encoder.stages.%d.layers.%d" % (si, bj)
            for k in range(3 if wide else 2):
                conv_bn("%s.layer.%d" % (h, k), "%s.conv%d" % (o, k + 1), "%s.bn%d" % (o, k + 1))
            if block.downsample is not None:
                conv_bn(h + ".shortcut", o + ".downsample.0", o + ".downsample.1")
    hf.load_state_dict(sd, strict=True)           # every tensor of the independent model is covered by the mapping
    for train in (True, False):
        enc.train(train); hf.train(train)
        x = torch.from_numpy(np.random.RandomState(7).rand(2, 3, 64, 96).astype(np.float32)).requires_grad_(True)
        feats = enc(x)
        x2 = x.detach().clone().requires_grad_(True)
        hs = hf((x2 - 0.45) / 0.225, output_hidden_states=True).hidden_states
        assert len(hs) == 5
        assert_close(npy(trunk.maxpool(feats[0])), npy(hs[0]), rtol=1e-5, atol=1e-6, what="stem + max-pool (train=%s)" % train)
        for i in range(1, 5):
            assert_close(npy(feats[i]), npy(hs[i]), rtol=1e-4, atol=1e-5, what="stage %d (train=%s)" % (i, train))
        w = torch.from_numpy(np.random.RandomState(8).randn(*feats[4].shape).astype(np.float32))
        (gx,) = torch.autograd.grad((feats[4] * w).sum() + feats[2].sum(), x)
        (gx2,) = torch.autograd.grad((hs[4] * w).sum() + hs[2].sum(), x2)
        assert_close(npy(gx), npy(gx2), rtol=1e-3, atol=1e-5 * float(gx2.abs().max()), what="input gradient (train=%s)" % train)


def test_derived_hparams_match_reference_rules():
    hp = OT.derived_hparams(12)
    assert (hp.accumulate_step, hp.micro_batch, hp.scheduler_step_size, hp.num_epochs) == (2, 6, 6, 11)
    assert abs(hp.learning_rate - 1.5e-4) < 1e-12
    hp = OT.derived_hparams(5)
    assert (hp.accumulate_step, hp.micro_batch) == (1, 5)


def test_depth_monitoring_metrics_match_reference(golden):
    """oracle.trainer.compute_depth_losses == the reference's Trainer.compute_depth_losses (trainer.py:598-630)."""
    from oracle import trainer as OT
    g = golden("depth_losses_b2_192x640")
    gt, pred = gin.depth_eval_inputs(int(g["seed"]), 2, 192, 640)
    got = OT.compute_depth_losses(torch.from_numpy(pred), torch.from_numpy(gt))
    np.testing.assert_allclose(got, g["metrics"], rtol=1e-6, atol=0)


def test_refiner_step_matches_reference(golden):
    """oracle.refiner.process_batch (refine inputs, refine decoder, warp, photometric + GDC SI-log loss) == the reference's
    Refiner.process_batch / compute_losses (refiner.py:299-382, 592-693) on the same weights, inputs and noise."""
    from oracle import refiner as OR
    g = golden("refiner_b1_192x640")
    B, H, W = 1, 192, 640
    opt = OR.default_opt(batch_size=B)
    models = gin.refiner_models(OR.build_models(opt, 0))
    inp, _ = gin.refiner_inputs(808, B, H, W)
    torch.manual_seed(int(g["noise_seed"]))
    noise = [torch.randn(B, 2, H, W) for _ in opt.scales]
    np.testing.assert_array_equal(noise[0].numpy().reshape(-1)[:16], g["noise_head"])
    outputs, losses = OR.process_batch(opt, models, inp, noise)
    for k, v in losses.items():
        np.testing.assert_allclose(float(v), float(g["L/" + k.replace("/", "_")]), rtol=2e-6, atol=1e-9, err_msg=k)
    for s in opt.scales:
        np.testing.assert_allclose(outputs[("disp", s)].detach().numpy(), g["disp%d" % s], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(outputs[("depth", 0, s)].detach().numpy()[:, :, ::16, ::16], g["depth%d_sub" % s], rtol=1e-5,
                                   atol=1e-5)
    params = dict(models["refine2d_decoder"].named_parameters())
    grads = torch.autograd.grad(losses["loss"], list(params.values()), allow_unused=True)
    for (n, _), gr in zip(params.items(), grads):
        if gr is not None:
            check_grad_compact(g, "g/" + n, gr, rtol=2e-4)



def test_rasterize_matches_reference(golden):
    """oracle.rasterize (kitti_utils.generate_depth_map + get_4beam restated) == the reference's output, bit for bit."""
    from oracle import rasterize as OR
    g = golden("rasterize_scan3")
    velo, P = gin.lidar_scan(int(g["seed"]))
    full = OR.depth_image(velo, P, 375, 1242)
    want = np.zeros((375, 1242))
    want[g["full_rows"], g["full_cols"]] = g["full_vals"]
    assert np.array_equal(full, want)
    # the scan exercises both duplicate rules: pixels hit more than once, and the (r, 0) / (r - 1, W - 1) index collision
    col, row, _ = OR.project_points(velo, P, 375, 1242)
    pix = row * 1242 + col
    assert np.unique(pix).size < pix.size
    assert ((col == 0) & (row == 101)).any() and ((col == 1241) & (row == 100)).any()
    beam = OR.four_beam(velo, P, 375, 1242)
    assert beam.dtype == np.float32 and beam.shape == (192, 640)
    assert np.array_equal(beam, g["beam"])
    # vel_depth=True (forward distance instead of camera z) and the crop branch of the padding (target shorter than the image)
    want = np.zeros((375, 1242))
    want[g["vd_rows"], g["vd_cols"]] = g["vd_vals"]
    assert np.array_equal(OR.depth_image(velo, P, 375, 1242, vel_depth=True), want)
    want = np.zeros(tuple(g["crop_shape"]))
    want[g["crop_rows"], g["crop_cols"]] = g["crop_vals"]
    assert np.array_equal(OR.pad_to_shape(full, (352, 1280)), want)


def test_rasterize_edge_cases():
    from oracle import rasterize as OR
    _, P = gin.lidar_scan(3)
    empty = np.zeros((0, 4), np.float32)
    assert not OR.depth_image(empty, P, 375, 1242).any()
    behind = np.array([[-5.0, 0.0, 0.0, 0.0]], np.float32)
    assert not OR.depth_image(behind, P, 375, 1242).any()
    # crop branch (target shorter than the image): 2 rows dropped after top padding (kitti_utils.py:97-99)
    d = OR.pad_to_shape(np.ones((375, 1242)), (352, 1280))
    assert d.shape == (375 + 23 - 2, 1280)


def _completor_case(seed, B, H, W, **over):
    from oracle import completor as OC
    opt = OC.default_opt(height=H, width=W, batch_size=B, **over)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp = gin.disp_pyramid(rng, B, H, W)
    outputs, leaves = {}, []
    for s in range(4):
        outputs[("disp", s)] = disp[("disp", s)].clone().requires_grad_(True)
        leaves.append(outputs[("disp", s)])
    for f in (-1, 1):
        aa, tr = gin.small_poses(rng, B)
        outputs[("cam_T_cam", 0, f)] = OL.transformation_from_parameters(aa, tr, invert=(f < 0)).detach()
    torch.manual_seed(1000 + seed)
    noise = [torch.randn(B, 2, H, W) for _ in range(4)]
    OT.generate_images_pred(opt, inp, outputs)
    losses = OC.compute_losses(opt, inp, outputs, noise)
    return opt, losses, torch.autograd.grad(losses["loss"], leaves)


def check_stored_grad(g, key, got, rtol, atol_rel):
    """Compare against a gradient stored by make_golden.put_grad (whole, or sum / L2 / every-97th sample)."""
    got = np.asarray(got)
    if key in g:
        assert_close(got, g[key], rtol=rtol, atol=atol_rel * np.abs(g[key]).max(), what=key)
        return
    s97 = g[key + "@s97"]
    assert_close(got.reshape(-1)[::97], s97, rtol=rtol, atol=atol_rel * np.abs(s97).max(), what=key + " sample")
    assert_close(np.sqrt((got.astype(np.float64) ** 2).sum()), g[key + "@l2"], rtol=max(rtol, 1e-5), atol=0, what=key + " L2")


@pytest.mark.parametrize("tag,over", [("default", {}), ("allscale", dict(completion_siloss_all_scale="true")),
                                      ("l1", dict(completion_siloss=False, completion_l1loss=True))])
def test_completor_losses_match_reference(golden, tag, over):
    """oracle.completor == reference Completor.generate_images_pred + compute_losses at 1216x352 (SI-log at scale 0, the
    all-scale switch, the masked-L1 alternative)."""
    g = golden("completor_b1_352x1216")
    opt, losses, grads = _completor_case(int(g["seed"]), 1, 352, 1216, **over)
    keys = {k[len(tag) + 3:] for k in g if k.startswith(tag + "/L/")}
    assert keys == {k.replace("/", "_") for k in losses}
    for k, v in losses.items():
        assert_close(float(v), g[tag + "/L/" + k.replace("/", "_")], rtol=2e-6, atol=1e-8, what=k)
    for s in range(4):
        check_stored_grad(g, tag + "/g_disp%d" % s, npy(grads[s]), 1e-5, 1e-6)


def test_completor_depth_metrics_match_reference(golden):
    from oracle import completor as OC
    g = golden("completor_b1_352x1216")
    gt, pred = gin.depth_eval_inputs(809, 2, 352, 1216, gt_h=352, gt_w=1216)
    for tag, crop in (("nocrop", False), ("crop", True)):
        m = OC.compute_depth_losses(OC.default_opt(completion_eigen_crop=crop), torch.from_numpy(pred), torch.from_numpy(gt))
        assert_close(m, g["metrics_" + tag], rtol=1e-5, atol=0, what="completor metrics " + tag)


def test_evaluate_metric_core_matches_reference(golden):
    """oracle.evaluate.compute_errors / batch_post_process_disparity == the reference's own functions (evaluate_depth.py:42-70)."""
    from oracle import evaluate as OE
    g = golden("evaluate_metrics")
    for i, (gt, pred) in enumerate(gin.eval_pairs(int(g["seed"]))):
        assert_close(np.array(OE.compute_errors(gt, pred), dtype=np.float64), g["errors%d" % i], rtol=1e-7, atol=0, what="errors%d" % i)
    l, r = gin.disp_pair(4243, 2, 192, 640)
    pp = OE.batch_post_process_disparity(l, r)
    assert np.array_equal(pp[:, ::7, ::3], g["pp_sub"])
    assert np.array_equal(np.concatenate([pp[:, :, :40], pp[:, :, -40:]], 2)[:, ::16], g["pp_edges"])
    # the resize restatement: identity at equal size, exact on a linear ramp in the interior, edge replication outside
    ramp = np.tile(np.arange(8, dtype=np.float32), (4, 1))
    assert np.array_equal(OE.resize_bilinear(ramp, 4, 8), ramp)
    up = OE.resize_bilinear(ramp, 8, 16)
    assert np.allclose(up[0, 1:-1], (np.arange(16)[1:-1] + 0.5) / 2 - 0.5) and up[0, 0] == 0 and up[0, -1] == 7


def test_absrel_fixtures_are_consistent(golden):
    """The paired AbsRel fixture (tests/golden/make_absrel_paired.py: 12 streams x {oracle, oracle one ulp away}, generated on 4
    threads) beside the distribution fixture of round 4 (make_absrel_stat.py, 16 threads): every run starts from ONE initial state
    (equal to 1e-6 across the fixtures), and the one-ulp partner differs from its base from the first checkpoint on (it IS another
    trajectory).  Streams 0-5 have the same seeds in both fixtures, and the same oracle code produced them - yet their AbsRel differ
    by up to 0.45 after 20 steps: torch's CPU kernels sum in a thread-count dependent order, which is a one-ulp perturbation of its
    own.  The reference's arithmetic is not reproducible against ITSELF at the 0.001 level beyond the first steps; no assertion can
    be made about that except that it is so (recorded here so that nobody tightens the GPU test's bound by mistake)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_absrel_paired as MP
    import make_absrel_stat as MS
    p, s = golden(MP.NAME), golden(MS.NAME)
    base, ulp = p["base"][:, :, 0], p["ulp"][:, :, 0]
    assert base.shape == ulp.shape == (MP.K, len(MS.CHECK)) and list(p["check"]) == list(MS.CHECK)
    assert np.allclose(base[:, 0], base[0, 0], rtol=0, atol=1e-7) and np.allclose(ulp[:, 0], base[:, 0], rtol=0, atol=2e-6)
    assert np.allclose(s["metrics"][:, 0, 0], base[0, 0], rtol=0, atol=1e-6)
    assert (np.abs(ulp[:, 1:] - base[:, 1:]) > 1e-4).all()
    other_threads = np.abs(s["metrics"][:, 1:, 0] - base[:MS.K, 1:])
    assert other_threads.max() > 1e-3, "the oracle became thread-count independent: tighten test_absrel_paired_gap_vs_oracle"
