"""HIP-backed Completor (fusiondepth_amd/completor.py) against the reference golden and the CPU oracle."""
import numpy as np
import pytest
import torch

import inputs as gin
from conftest import assert_close, assert_mostly_close
from oracle import completor as OC
from oracle import layers as OL
from oracle import trainer as OT
from test_oracle_golden import check_stored_grad

pytestmark = pytest.mark.gpu


def _opts(*flags, **over):
    from fusiondepth_amd.options import MonodepthOptions
    o = MonodepthOptions().parse(["--weights_init", "scratch", "--batch_size", "1", "--completion_num_layers", "18"] + list(flags))
    for k, v in over.items():
        setattr(o, k, v)
    return o


@pytest.mark.parametrize("tag,flags", [("default", []), ("allscale", ["--completion_siloss_all_scale", "true"]),
                                       ("l1", ["--completion_siloss", "--completion_l1loss"])])
def test_completor_loss_path_vs_reference_golden(golden, tag, flags):
    """Completor.generate_images_pred + compute_losses (fused kernel, si_mode 0 / 1, scale-0-only LiDAR term) at 1216x352 on
    the inputs of tests/golden/make_golden.py::gold_completor -> the losses and disparity gradients the REFERENCE produced."""
    from fusiondepth_amd.completor import Completor
    g = golden("completor_b1_352x1216")
    seed, B, H, W = int(g["seed"]), 1, 352, 1216
    opt = _opts(*flags)
    cp = Completor(opt, verbose=False)
    assert (opt.height, opt.width) == (H, W)                       # completor.py:31-34 forces the completion resolution
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp = gin.disp_pyramid(rng, B, H, W)
    ginp = {k: v.cuda() for k, v in inp.items()}
    outputs, leaves = {}, []
    for s in range(4):
        outputs[("disp", s)] = disp[("disp", s)].cuda().requires_grad_(True)
        leaves.append(outputs[("disp", s)])
    for f in (-1, 1):
        aa, tr = gin.small_poses(rng, B)
        outputs[("cam_T_cam", 0, f)] = OL.transformation_from_parameters(aa, tr, invert=(f < 0)).cuda()
    torch.manual_seed(1000 + seed)
    ginp["_noise"] = [torch.randn(B, 2, H, W).cuda() for _ in range(4)]
    w0 = opt.completion_siloss_weight
    cp.generate_images_pred(ginp, outputs, opt.frame_ids)
    losses = cp.compute_losses(ginp, outputs)
    keys = {k[len(tag) + 3:] for k in g if k.startswith(tag + "/L/")}
    assert keys == {k.replace("/", "_") for k in losses}
    for k, v in losses.items():
        assert_close(float(v.detach()), g[tag + "/L/" + k.replace("/", "_")], rtol=1e-4, atol=1e-7, what=k)
    assert opt.completion_siloss_weight == float(g[tag + "/siloss_weight_after"]) == (w0 if tag == "allscale" else 2 * w0)
    grads = torch.autograd.grad(losses["loss"], leaves)
    for s in range(4):
        key = tag + "/g_disp%d" % s
        got = grads[s].cpu().numpy()
        ref = g[key] if key in g else g[key + "@s97"]
        cmp_ = got if key in g else got.reshape(-1)[::97]
        # argmin / clamp ties flip single pixels: per-entry tolerance for all but 1 % (2.5 % at the coarsest scale,
        # where one entry collects 16x16 pixels), aggregate L1 bound 1e-3.  The golden is the reference's float32 run, which is
        # itself 1.2 % / 9e-4 away from a float64 evaluation at this size (scripts/debug_ms64.py); the kernel is held to the
        # float64 ground truth in test_gpu_losspath.py::test_loss_path_error_against_float64.
        assert_mostly_close(cmp_, ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max(), what=key, max_bad_frac=2.5e-2 if s == 3 else 1e-2)
        if key + "@l2" in g:
            assert_close(np.sqrt((got.astype(np.float64) ** 2).sum()), g[key + "@l2"], rtol=2e-3, atol=0, what=key + " L2")


def test_completor_metrics_vs_reference_golden(golden):
    from fusiondepth_amd.completor import Completor
    g = golden("completor_b1_352x1216")
    gt, pred = gin.depth_eval_inputs(809, 2, 352, 1216, gt_h=352, gt_w=1216)
    for tag, flags in (("nocrop", []), ("crop", ["--completion_eigen_crop"])):
        cp = Completor(_opts(*flags), verbose=False)
        m = {}
        cp.compute_depth_losses({"depth_gt": torch.from_numpy(gt).cuda()}, {("depth", 0, 0): torch.from_numpy(pred).cuda()}, m)
        assert_close([float(m[n]) for n in cp.depth_metric_names], g["metrics_" + tag], rtol=2e-5, atol=0, what="metrics " + tag)


def _pair(opt, seed=5):
    from fusiondepth_amd.completor import Completor
    cp = Completor(opt, verbose=False)
    oopt = OC.default_opt(height=opt.height, width=opt.width, batch_size=opt.batch_size,
                          completion_num_layers=opt.completion_num_layers,
                          completion_pose_num_layers=opt.completion_pose_num_layers, completion_siloss=opt.completion_siloss,
                          completion_l1loss=opt.completion_l1loss, completion_siloss_all_scale=opt.completion_siloss_all_scale)
    om = OC.build_models(oopt, seed)
    for k, m in om.items():
        gin.fill_params(m, 100 + len(k))
        with torch.no_grad():
            for name, t in cp.models[k].state_dict().items():
                t.copy_(m.state_dict()[name])
    return cp, oopt, om


@pytest.mark.parametrize("layers,flags", [(50, []), (18, ["--completion_siloss", "--completion_l1loss"])], ids=["r50-si", "r18-l1"])
def test_completor_train_steps_match_oracle(layers, flags):
    """Two optimiser steps of the completion driver (completor.py:229-246: one Adam step per batch at the plain learning rate,
    ResNet-50 colour / beam encoders + ResNet-18 pose encoders by default) against the oracle with torch.optim.Adam."""
    from test_gpu_trainer import _batch
    B, H, W = 2, 64, 96
    opt = _opts("--completion_not_full_res", "--height", str(H), "--width", str(W), *flags, batch_size=B,
                completion_num_layers=layers)
    cp, oopt, om = _pair(opt)
    assert cp.accumulate_step == 1 and cp.lr == opt.learning_rate and cp.scheduler_step_size == 25 and cp.opt.num_epochs == 3
    for m in om.values():
        m.train()
    adam = torch.optim.Adam(OT.trainable_parameters(om), oopt.learning_rate)
    seq = []
    for step in range(2):
        inp, noise = _batch(B, H, W, 900 + step)
        ginp = {k: v.cuda() for k, v in inp.items()}
        ginp["_noise"] = [n.cuda() for n in noise]
        _, lo = OC.process_batch(oopt, om, inp, noise)
        adam.zero_grad()
        lo["loss"].backward()
        adam.step()
        lg = cp.train_step([ginp])
        assert set(lg) == set(lo)
        seq.append((float(lg["loss"]), float(lo["loss"])))
        if step == 0:
            for k in lo:
                assert_close(float(lg[k]), float(lo[k]), rtol=5e-4, atol=1e-6, what="step0 " + k)
    print("completor loss (HIP, oracle):", seq)
    assert_close(seq[1][0], seq[1][1], rtol=1e-2, atol=0, what="loss after one optimiser step")


def test_completor_val_best_rms_bookkeeping(tmp_path):
    """completor.py:390-426: eval-mode forward without the pose nets, mean metrics, best de/rms, checkpoint named rms<N>."""
    from test_gpu_trainer import _batch
    B, H, W = 1, 64, 96
    opt = _opts("--completion_not_full_res", "--height", str(H), "--width", str(W), "--log_dir", str(tmp_path), batch_size=B)
    cp, oopt, om = _pair(opt)
    batches = []
    for i in range(2):
        inp, _ = _batch(B, H, W, 950 + i)
        gt, _p = gin.depth_eval_inputs(960 + i, B, H, W, gt_h=H, gt_w=W)
        inp["depth_gt"] = torch.from_numpy(gt / 70.0)                       # sub-metre so that rms (mm) < 1200
        batches.append({k: v.cuda() for k, v in inp.items()})
    losses, saved = cp.val(batches)
    # oracle: eval-mode nets, metrics per batch, mean over batches
    for m in om.values():
        m.eval()
    want = np.zeros(7)
    with torch.no_grad():
        for b in batches:
            cb = {k: v.cpu() for k, v in b.items()}
            outs = om["depth"](om["encoder"](cb[("color_aug", 0, 0)]), beam_features=om["beam_encoder"](cb["2channel"]))
            disp = torch.nn.functional.interpolate(outs[("disp", 0)], [H, W], mode="bilinear", align_corners=False)
            depth = OL.disp_to_depth(disp, oopt.min_depth, oopt.max_depth)[1]
            want += np.array(OC.compute_depth_losses(oopt, depth, cb["depth_gt"]))
    want /= len(batches)
    assert_close([float(losses[n]) for n in cp.depth_metric_names], want, rtol=2e-3, atol=0, what="val metrics")
    assert cp.best == float(losses["de/rms"])
    rms = round(float(losses["de/rms"]))
    assert (saved is not None) == (rms < 1200)
    if saved is not None:
        assert saved.endswith("weights_rms%d" % rms)
    assert all(m.training for m in cp.models.values())                     # set_train() restored
    again, saved2 = cp.val(batches)                                        # not better than itself: no new checkpoint
    assert saved2 is None


def _sparse_batch(B, H, W, seed, n_points):
    """test_gpu_trainer._batch with r100 / r200 LiDAR maps (n_points random ROI pixels per image, 3.5 - 7 m) instead of 4 scan lines"""
    from oracle import scatter as OS
    from fusiondepth_amd import functional as FD
    inp, _ = gin.batch_inputs(seed, B, H, W)
    roi = FD.scaled_roi(H, W)
    for i, f in enumerate((0, -1, 1)):
        beam = gin.lidar_random(np.random.RandomState(seed + 30 + i), B, H, W, n_points, roi, lo=3.5, hi=7.0)
        two = np.stack([np.stack(OS.scatter_2channel_c(beam[b, 0], roi)) for b in range(B)])
        inp[("2channel", f, 0)] = torch.from_numpy(two)
        if f == 0:
            inp["2channel"] = torch.from_numpy(two)
            inp["4beam"] = torch.from_numpy(beam)
    noise = [torch.from_numpy(np.random.RandomState(seed + 50 + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
    return inp, noise


@pytest.mark.parametrize("n_points,flags", [(100, []), (200, ["--completion_siloss", "--completion_l1loss"])], ids=["r100-si", "r200-l1"])
def test_completor_step_on_random_sample_lidar_vs_oracle(n_points, flags):
    """BASELINE config 5's sparse inputs through a Completor step ("missing" 4): r100 / r200 maps at 640x192 - the LiDAR
    term's masked reductions (completor.py:621-725: SI-log or masked L1) run on at most n_points pixels per image, the beam encoders see a
    nearly empty 2-channel map.  One process_batch + backward: every loss against the oracle, the parameter gradient of every network
    in relative L2 (float32 oracle; train-mode BatchNorm over 2 images: the bound is the one of the dense-LiDAR trainer tests)."""
    import conftest
    B, H, W = 2, 192, 640
    opt = _opts("--completion_not_full_res", "--height", str(H), "--width", str(W), *flags, batch_size=B, completion_num_layers=18)
    cp, oopt, om = _pair(opt)
    for m in om.values():
        m.train()
    inp, noise = _sparse_batch(B, H, W, 940 + n_points, n_points)
    assert int((inp["4beam"] > 0).sum()) == B * n_points
    ginp = {k: v.cuda() for k, v in inp.items()}
    ginp["_noise"] = [n.cuda() for n in noise]
    for m in om.values():
        for p_ in m.parameters():
            p_.grad = None
    _, lo = OC.process_batch(oopt, om, inp, noise)
    lo["loss"].backward()
    cp.flat.zero_grad()
    _, lg = cp.process_batch(ginp)
    assert set(lg) == set(lo)
    for k in lo:
        a, b = float(lg[k].detach()), float(lo[k])
        assert np.isnan(a) == np.isnan(b), "%s: HIP %r, oracle %r" % (k, a, b)
        if not np.isnan(b):
            conftest.report("completor %d-point LiDAR: %s |HIP - oracle| / |oracle|" % (n_points, k), abs(a - b) / max(abs(b), 1e-30), 5e-4)
            assert_close(a, b, rtol=5e-4, atol=1e-6, what=k)
    lg["loss"].backward()
    cp._join_side_streams()
    torch.cuda.synchronize()
    for name, m in cp.models.items():
        got = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).reshape(-1) for p_ in m.parameters()]).double().cpu()
        want = torch.cat([(p_.grad if p_.grad is not None else torch.zeros_like(p_)).reshape(-1) for p_ in om[name].parameters()]).double()
        if float(want.norm()) == 0.0:
            assert float(got.norm()) == 0.0, name
            continue
        e = float((got - want).norm() / want.norm())
        conftest.report("completor %d-point LiDAR: %s parameter gradient, relative L2 vs the float32 oracle" % (n_points, name), e, 5e-2)
        assert e <= 5e-2, "%s: %.3g" % (name, e)
