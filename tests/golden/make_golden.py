#!/usr/bin/env python
"""Generate golden vectors by IMPORTING AND RUNNING THE REFERENCE in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Runs only where /root/reference exists (never on the GPU box).  The reference has no tests or
known-answer vectors of its own, so these outputs *are* the pin for ``oracle/``.
Only arithmetic-free third-party imports are stubbed (tensorboardX, wandb, cv2, skimage,
torchvision — the latter only so that ``import networks``/``import trainer`` succeed; no torchvision
arithmetic is used by anything captured here).  Inputs come from ``inputs.py`` (seeded numpy) and
are NOT stored, except small ones; outputs and autograd gradients are.
"""
import ast
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FD_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))     # repo root: the oracle package
import inputs as gin  # noqa: E402


def _stub_modules():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("tensorboardX", SummaryWriter=object)
    mod("wandb")
    mod("cv2")
    sk = mod("skimage")
    sk.transform = mod("skimage.transform")
    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms")

    class _ResNetPlaceholder(nn.Module):
        pass

    res = mod("torchvision.models.resnet", BasicBlock=object, Bottleneck=object, model_urls={})
    tv.models = mod("torchvision.models", ResNet=_ResNetPlaceholder, resnet=res)
    for n in ("resnet18", "resnet34", "resnet50", "resnet101", "resnet152"):
        setattr(tv.models, n, None)


def load_reference():
    _stub_modules()
    sys.path.insert(0, REF)
    sys.argv = ["trainer.py", "--no_cuda"]
    torch.Tensor.cuda = lambda self, *a, **k: self      # trainer.py:551-552 hard-codes .cuda()
    import layers as ref_layers
    import trainer as ref_trainer

    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    dd = by_path("ref_depth_decoder", "networks/depth_decoder.py")
    pd = by_path("ref_pose_decoder", "networks/pose_decoder.py")
    pc = by_path("ref_pose_cnn", "networks/pose_cnn.py")
    src = open(os.path.join(REF, "gen2channel.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "get_4beam_2channel"][0]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "gen2channel.py", "exec"), ns)
    return ref_layers, ref_trainer, dd, pd, pc, ns["get_4beam_2channel"]


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %8.1f KiB  (%d arrays)" % (name + ".npz", os.path.getsize(path) / 1024, len(arrays)))


def put_grad(out, key, g):
    """Small gradients are stored whole; large ones as (sum, L2, every-97th-element sample)."""
    a = npy(g)
    if a.size <= 20000:
        out[key] = a
    else:
        out[key + "@sum"] = np.array(a.astype(np.float64).sum())
        out[key + "@l2"] = np.array(np.sqrt((a.astype(np.float64) ** 2).sum()))
        out[key + "@s97"] = a.reshape(-1)[::97].copy()


# ------------------------------------------------------------------------------------------------
def gold_layers(RL):
    B, H, W = 2, 32, 64
    inp, rng = gin.batch_inputs(101, B, H, W)
    out = {}
    disp = gin.disp_pyramid(rng, B, H, W)[("disp", 0)].requires_grad_(True)
    sd, depth = RL.disp_to_depth(disp, 0.1, 100.0)
    out["d2d_scaled"], out["d2d_depth"] = npy(sd), npy(depth)

    aa, tr = gin.small_poses(rng, B)
    aa.requires_grad_(True), tr.requires_grad_(True)
    cot_T = torch.from_numpy(rng.randn(B, 4, 4).astype(np.float32))
    for inv in (False, True):
        M = RL.transformation_from_parameters(aa, tr, invert=inv)
        g = torch.autograd.grad((M * cot_T).sum(), [aa, tr])
        tag = "inv" if inv else "fwd"
        out["T_" + tag], out["T_%s_gaa" % tag], out["T_%s_gtr" % tag] = npy(M), npy(g[0]), npy(g[1])
    out["T_cot"] = npy(cot_T)

    K, inv_K = inp[("K", 0)], inp[("inv_K", 0)]
    T = RL.transformation_from_parameters(aa, tr, invert=False).detach().requires_grad_(True)
    bp, pj = RL.BackprojectDepth(B, H, W), RL.Project3D(B, H, W)
    depth_in = depth.detach().clone().requires_grad_(True)
    pts = bp(depth_in, inv_K)
    grid = pj(pts, K, T)
    cot_g = torch.from_numpy(rng.randn(B, H, W, 2).astype(np.float32))
    g = torch.autograd.grad((grid * cot_g).sum(), [depth_in, T])
    out.update(bp_points=npy(pts), pj_grid=npy(grid), pj_cot=npy(cot_g), pj_gdepth=npy(g[0]), pj_gT=npy(g[1]))
    out["pj_T"] = npy(T)

    x = inp[("color", 0, 0)].clone().requires_grad_(True)
    y = inp[("color", 1, 0)].clone().requires_grad_(True)
    s = RL.SSIM()(x, y)
    cot_s = torch.from_numpy(rng.rand(B, 3, H, W).astype(np.float32))
    g = torch.autograd.grad((s * cot_s).sum(), [x, y])
    out.update(ssim=npy(s), ssim_cot=npy(cot_s), ssim_gx=npy(g[0]), ssim_gy=npy(g[1]))

    d = disp.detach().clone().requires_grad_(True)
    sm = RL.get_smooth_loss(d, inp[("color", 0, 0)])
    out["smooth"], out["smooth_gdisp"] = npy(sm), npy(torch.autograd.grad(sm, d)[0])

    torch.manual_seed(7)
    cb = RL.ConvBlock(5, 7)
    c3 = RL.Conv3x3(5, 1, use_refl=False)
    xin = torch.from_numpy(rng.randn(B, 5, H, W).astype(np.float32)).requires_grad_(True)
    yo = cb(xin)
    cot_c = torch.from_numpy(rng.randn(*yo.shape).astype(np.float32))
    g = torch.autograd.grad((yo * cot_c).sum(), [xin, cb.conv.conv.weight, cb.conv.conv.bias])
    out.update(cb_x=npy(xin), cb_w=npy(cb.conv.conv.weight), cb_b=npy(cb.conv.conv.bias), cb_y=npy(yo),
               cb_cot=npy(cot_c), cb_gx=npy(g[0]), cb_gw=npy(g[1]), cb_gb=npy(g[2]))
    out.update(c3_w=npy(c3.conv.weight), c3_b=npy(c3.conv.bias), c3_y=npy(c3(xin)))
    out["up_y"] = npy(RL.upsample(xin))

    cxy = RL.Cat_xy(B, H, W)(depth.detach().clone(), inv_K)
    out["catxy"] = npy(cxy)

    gt = torch.from_numpy(rng.uniform(1, 80, size=5000).astype(np.float32))
    pr = torch.from_numpy(rng.uniform(1, 80, size=5000).astype(np.float32))
    out["errs_gt"], out["errs_pred"] = npy(gt), npy(pr)
    out["errs"] = np.array([float(v) for v in RL.compute_depth_errors(gt, pr)], dtype=np.float64)
    save("layers_b2_32x64", **out)


def gold_decoders(DD, PD, PC):
    B, H, W = 2, 64, 96
    rng = np.random.RandomState(202)
    ch = np.array([64, 64, 128, 256, 512])
    feats, beams = gin.feature_pyramids(rng, B, H, W, ch)
    for t in feats + beams:
        t.requires_grad_(True)
    out = {}

    dec = DD.DepthDecoder(ch)
    gin.fill_params(dec, 11)
    o = dec(feats, beam_features=beams)
    cots = {s: torch.from_numpy(rng.randn(*o[("disp", s)].shape).astype(np.float32)) for s in range(4)}
    loss = sum((o[("disp", s)] * cots[s]).sum() for s in range(4))
    params = list(dec.parameters())
    g = torch.autograd.grad(loss, feats + beams + params)
    for s in range(4):
        out["dec_disp%d" % s], out["dec_cot%d" % s] = npy(o[("disp", s)]), npy(cots[s])
    for i in range(5):
        put_grad(out, "dec_gfeat%d" % i, g[i])
        put_grad(out, "dec_gbeam%d" % i, g[5 + i])
    for (k, _), gv in zip(dec.named_parameters(), g[10:]):
        put_grad(out, "dec_g/" + k.replace(".", "/"), gv)
    # no-beam forward as well (depth_decoder.py:71-72,79-80)
    o2 = dec([f.detach() for f in feats])
    out["dec_nobeam_disp0"] = npy(o2[("disp", 0)])

    pose = PD.PoseDecoder(ch, num_input_features=1, num_frames_to_predict_for=2)
    gin.fill_params(pose, 12)
    f4 = feats[4].detach().clone().requires_grad_(True)
    b4 = beams[4].detach().clone().requires_grad_(True)
    aa, tr = pose([[None, None, None, None, f4]], beam_inputs=[[None, None, None, None, b4]])
    cot_a = torch.from_numpy(rng.randn(*aa.shape).astype(np.float32))
    cot_t = torch.from_numpy(rng.randn(*tr.shape).astype(np.float32))
    g = torch.autograd.grad((aa * cot_a).sum() + (tr * cot_t).sum(), [f4, b4] + list(pose.parameters()))
    out.update(pose_aa=npy(aa), pose_tr=npy(tr), pose_cot_a=npy(cot_a), pose_cot_t=npy(cot_t),
               pose_gf4=npy(g[0]), pose_gb4=npy(g[1]))
    for (k, _), gv in zip(pose.named_parameters(), g[2:]):
        put_grad(out, "pose_g/" + k.replace(".", "/"), gv)

    pcnn = PC.PoseCNN(2)
    gin.fill_params(pcnn, 13)
    xin = torch.from_numpy(np.random.RandomState(213).rand(B, 6, H, W).astype(np.float32))
    a2, t2 = pcnn(xin)
    out.update(posecnn_aa=npy(a2), posecnn_tr=npy(t2))
    save("decoders_b2_64x96", **out)

    # refiner variant of the decoder (road/catxy/deep + depth_maps + tanh; depth_decoder.py:39-42,81-90)
    dec2 = DD.DepthDecoder(ch, road=True, catxy=True, deep=True)
    gin.fill_params(dec2, 14)
    out2 = {}
    rng2 = np.random.RandomState(214)
    dm = {("disp", s): torch.from_numpy(rng2.rand(B, 6, H // 2 ** s, W // 2 ** s).astype(np.float32)) for s in range(4)}
    o3 = dec2([f.detach() for f in feats], beam_features=[b.detach() for b in beams], depth_maps=dm, tanh=True)
    for s in range(4):
        out2["disp%d" % s] = npy(o3[("disp", s)])
    out2["num_params"] = np.array(sum(p.numel() for p in dec2.parameters()))
    save("decoder_refine_b2_64x96", **out2)


def _trainer_self(RL, RT, opt_over, B, H, W):
    from types import SimpleNamespace
    import copy
    opt = copy.deepcopy(RT.opts)
    opt.height, opt.width = H, W
    for k, v in opt_over.items():
        setattr(opt, k, v)
    ns = SimpleNamespace(opt=opt, batch_size=B, num_scales=len(opt.scales), ssim=RL.SSIM(),
                         backproject_depth={}, project_3d={})
    for s in opt.scales:
        ns.backproject_depth[s] = RL.BackprojectDepth(B, H // 2 ** s, W // 2 ** s)
        ns.project_3d[s] = RL.Project3D(B, H // 2 ** s, W // 2 ** s)
    ns.compute_reprojection_loss = lambda pred, target: RT.Trainer.compute_reprojection_loss(ns, pred, target)
    return ns


def gold_losses(RL, RT, name, seed, B, H, W, full_arrays, opt_over=None, empty_si=None):
    """``empty_si``: make the LiDAR validity mask of trainer.py:580-584 empty - "all": no LiDAR returns at all; "scale2": the
    1/4-scale disparity predicts 2.9 m everywhere (no return within 2 m of it).  The reference then takes the mean of an empty
    selection: that scale's si_loss and the total are NaN, the gradients stay finite (trainer.py:585-589)."""
    ns = _trainer_self(RL, RT, opt_over or {}, B, H, W)
    inp, rng = gin.batch_inputs(seed, B, H, W)
    disp = gin.disp_pyramid(rng, B, H, W)
    gin.make_si_mask_empty(inp, disp, empty_si)
    outputs, leaves = {}, []
    for s in range(4):
        outputs[("disp", s)] = disp[("disp", s)].clone().requires_grad_(True)
        leaves.append(outputs[("disp", s)])
    Ts = {}
    for f in (-1, 1):
        aa, tr = gin.small_poses(rng, B)
        Ts[f] = RL.transformation_from_parameters(aa, tr, invert=(f < 0)).detach().requires_grad_(True)
        outputs[("cam_T_cam", 0, f)] = Ts[f]
        leaves.append(Ts[f])
    RT.Trainer.generate_images_pred(ns, inp, outputs, ns.opt.frame_ids)
    noise_seed = 1000 + seed
    torch.manual_seed(noise_seed)
    losses = RT.Trainer.compute_losses(ns, inp, outputs)
    torch.manual_seed(noise_seed)
    noise = [torch.randn(B, 2, H, W) for _ in ns.opt.scales]
    grads = torch.autograd.grad(losses["loss"], leaves)
    out = {"noise_seed": np.array(noise_seed)}
    for k, v in losses.items():
        out["L/" + k.replace("/", "_")] = np.array(float(v.detach()), dtype=np.float64)
    for s in range(4):
        out["g_disp%d" % s] = npy(grads[s])
        if "identity_selection/%d" % s in outputs:
            out["idsel%d" % s] = npy(outputs["identity_selection/%d" % s]).astype(np.uint8)
        if full_arrays:
            out["noise%d" % s] = npy(noise[s])
            out["depth%d" % s] = npy(outputs[("depth", 0, s)])
            for f in (-1, 1):
                out["sample%d_%d" % (f, s)] = npy(outputs[("sample", f, s)])
                out["color%d_%d" % (f, s)] = npy(outputs[("color", f, s)])
        else:   # full-size case: keep a strided subsample only
            out["depth%d_sub" % s] = npy(outputs[("depth", 0, s)])[:, :, ::16, ::16]
            for f in (-1, 1):
                out["color%d_%d_sub" % (f, s)] = npy(outputs[("color", f, s)])[:, :, ::16, ::16]
    out["g_T-1"], out["g_T1"] = npy(grads[4]), npy(grads[5])
    out["T-1"], out["T1"] = npy(Ts[-1]), npy(Ts[1])
    # first few noise values so the test can prove it regenerated the same stream
    out["noise_head"] = npy(noise[0]).reshape(-1)[:16]
    save(name, **out)


def gold_scatter(get2ch):
    rng = np.random.RandomState(303)
    beams = gin.lidar_4beam(rng, 2, 192, 640)[:, 0]
    third = np.zeros((192, 640), dtype=np.float32)            # dense clump: exercises equal-confidence averaging
    third[100:112, 300:330] = rng.uniform(0.05, 0.65, size=(12, 30)).astype(np.float32)
    third[80, 2] = 0.3
    third[189, 637] = 0.4
    third[76, 100:110:2] = 0.25
    maps = [beams[0], beams[1], third]
    out = {}
    for i, m in enumerate(maps):
        d, c = get2ch(m.astype(np.float64))   # the reference feeds float64 numpy (gen2channel.py:49-51)
        out["beam%d" % i], out["depth%d" % i], out["conf%d" % i] = m, npy(d), npy(c)
    save("scatter_192x640", **out)


def gold_depth_losses(RL, RT):
    """trainer.py:598-630 (Garg crop, median scaling, clamp, compute_depth_errors) run as an unbound Trainer method."""
    from types import SimpleNamespace
    B, H, W = 2, 192, 640
    gt, pred = gin.depth_eval_inputs(707, B, H, W)
    ns = SimpleNamespace(depth_metric_names=["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"])
    losses = {}
    RT.Trainer.compute_depth_losses(ns, {"depth_gt": torch.from_numpy(gt)}, {("depth", 0, 0): torch.from_numpy(pred)}, losses)
    save("depth_losses_b2_192x640", seed=np.array(707), metrics=np.array([float(losses[m]) for m in ns.depth_metric_names],
                                                                           dtype=np.float64))


def gold_refiner(RL, DD, PD):
    """refiner.py:299-382 + :592-693 run as unbound Refiner methods.  The ResNet trunks are oracle.networks' (torchvision is
    not installed); decoders, Cat_xy, projection, SSIM and every loss line are the reference's own code."""
    import copy
    from types import SimpleNamespace
    import refiner as RF
    from oracle import networks as ON
    B, H, W = 1, 192, 640
    opt = copy.deepcopy(RF.opts)
    opt.height, opt.width, opt.batch_size = H, W, B
    opt.refine_2d, opt.clone_gdc, opt.train_entire_net = True, True, False
    enc = ON.ResnetEncoder(18, False)
    models = {"encoder": enc, "beam_encoder": ON.ResnetEncoder(18, False, beam_encoder=True),
              "beam_encoder_pose": ON.ResnetEncoder(18, False, num_input_images=2, beam_encoder=True),
              "depth": DD.DepthDecoder(enc.num_ch_enc, opt.scales),
              "pose_encoder": ON.ResnetEncoder(18, False, num_input_images=2),
              "pose": PD.PoseDecoder(enc.num_ch_enc, num_input_features=1, num_frames_to_predict_for=2),
              "refine2d_decoder": DD.DepthDecoder(enc.num_ch_enc, opt.scales, road=True, catxy=(opt.catxy == "true"),
                                                  deep=(opt.refine2d_deep == "true"))}
    gin.refiner_models(models)
    ns = SimpleNamespace(opt=opt, models=models, device=torch.device("cpu"), batch_size=B, eval_scales=opt.scales,
                         num_scales=len(opt.scales), num_pose_frames=2, use_pose_net=True, ssim=RL.SSIM(),
                         backproject_depth={}, project_3d={}, catxy={})
    for s in opt.scales:
        h, w = H // 2 ** s, W // 2 ** s
        ns.backproject_depth[s] = RL.BackprojectDepth(B, h, w)
        ns.project_3d[s] = RL.Project3D(B, h, w)
        ns.catxy["False", s] = RL.Cat_xy(B, h, w)
    for name in ("predict_poses", "generate_images_pred", "compute_reprojection_loss", "siloss", "compute_losses"):
        setattr(ns, name, (lambda f: (lambda *a, **k: f(ns, *a, **k)))(getattr(RF.Refiner, name)))
    inp, noise = gin.refiner_inputs(808, B, H, W)
    torch.manual_seed(4242)
    rec = [torch.randn(B, 2, H, W) for _ in opt.scales]      # what compute_losses will draw (scale order)
    torch.manual_seed(4242)
    outputs, losses = RF.Refiner.process_batch(ns, {k: v.clone() for k, v in inp.items()})
    params = list(models["refine2d_decoder"].parameters())
    grads = torch.autograd.grad(losses["loss"], params, allow_unused=True)
    out = {"noise_seed": np.array(4242), "noise_head": npy(rec[0]).reshape(-1)[:16]}
    for k, v in losses.items():
        out["L/" + k.replace("/", "_")] = np.array(float(v.detach()) if torch.is_tensor(v) else float(v), dtype=np.float64)
    for s in opt.scales:
        out["disp%d" % s] = npy(outputs[("disp", s)])
        out["depth%d_sub" % s] = npy(outputs[("depth", 0, s)])[:, :, ::16, ::16]
    for (n, p), g in zip(models["refine2d_decoder"].named_parameters(), grads):
        if g is not None:
            put_grad(out, "g/" + n, g)
    save("refiner_b1_192x640", **out)


def gold_rasterize():
    """kitti_utils.generate_depth_map (:40-102) on a synthetic scan written as the reference's on-disk formats (Velodyne .bin,
    calib_cam_to_cam.txt / calib_velo_to_cam.txt), then kitti_dataset.get_4beam's pad + 2x2 ceil max-pool + / 100."""
    import tempfile
    import torch.nn.functional as F
    np.int = int                                     # kitti_utils.py:76 uses the alias removed in numpy 1.24
    import kitti_utils as KU
    velo, P = gin.lidar_scan(3)
    cal = gin.lidar_scan.calib
    fmt = lambda a: " ".join("%.17g" % v for v in np.asarray(a, dtype=np.float64).reshape(-1))
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "calib_cam_to_cam.txt"), "w") as f:
            f.write("S_rect_02: %s\nR_rect_00: %s\nP_rect_02: %s\n" % (fmt(cal["S_rect_02"]), fmt(cal["R_rect_00"]), fmt(cal["P_rect_02"])))
        with open(os.path.join(d, "calib_velo_to_cam.txt"), "w") as f:
            f.write("R: %s\nT: %s\n" % (fmt(cal["R"]), fmt(cal["T"])))
        velo.tofile(os.path.join(d, "scan.bin"))
        full = KU.generate_depth_map(d, os.path.join(d, "scan.bin"), 2)
        padded = KU.generate_depth_map(d, os.path.join(d, "scan.bin"), 2, shape=[384, 1280])
        vd = KU.generate_depth_map(d, os.path.join(d, "scan.bin"), 2, True)                      # vel_depth
        crop = KU.generate_depth_map(d, os.path.join(d, "scan.bin"), 2, shape=[352, 1280])       # shorter target: crop branch
    beam = F.max_pool2d(torch.tensor(padded).unsqueeze(0), 2, ceil_mode=True).squeeze().numpy()
    beam = torch.from_numpy(np.expand_dims(beam, 0).astype(np.float32)) / 100.0          # mono_dataset.py:196-198
    ys, xs = np.nonzero(full)
    vy, vx = np.nonzero(vd)
    cy, cx = np.nonzero(crop)
    save("rasterize_scan3", seed=np.array(3), full_rows=ys.astype(np.int32), full_cols=xs.astype(np.int32), full_vals=full[ys, xs],
         beam=beam.numpy()[0], vd_rows=vy.astype(np.int32), vd_cols=vx.astype(np.int32), vd_vals=vd[vy, vx],
         crop_shape=np.array(crop.shape), crop_rows=cy.astype(np.int32), crop_cols=cx.astype(np.int32), crop_vals=crop[cy, cx])


def gold_completor(RL):
    """completor.py:428-476 generate_images_pred + :546-726 compute_losses + :728-762 compute_depth_losses run as unbound
    Completor methods at the completion resolution (1216x352): default flags (SI-log at scale 0 only), the all-scale switch,
    and the masked-L1 alternative."""
    import copy
    from types import SimpleNamespace
    import completor as CP
    B, H, W = 1, 352, 1216
    names = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]
    out = {"seed": np.array(808)}
    for tag, over in (("default", {}), ("allscale", dict(completion_siloss_all_scale="true")),
                      ("l1", dict(completion_siloss=False, completion_l1loss=True))):
        opt = copy.deepcopy(CP.opts)
        opt.height, opt.width, opt.batch_size = H, W, B
        for k, v in over.items():
            setattr(opt, k, v)
        ns = SimpleNamespace(opt=opt, num_scales=len(opt.scales), ssim=RL.SSIM(), backproject_depth={}, project_3d={},
                             l1loss=nn.L1Loss(), depth_metric_names=names)
        for s in opt.scales:
            ns.backproject_depth[s] = RL.BackprojectDepth(B, H // 2 ** s, W // 2 ** s)
            ns.project_3d[s] = RL.Project3D(B, H // 2 ** s, W // 2 ** s)
        ns.compute_reprojection_loss = lambda pred, target, ns=ns: CP.Completor.compute_reprojection_loss(ns, pred, target)
        inp, rng = gin.batch_inputs(808, B, H, W)
        disp = gin.disp_pyramid(rng, B, H, W)
        outputs, leaves = {}, []
        for s in range(4):
            outputs[("disp", s)] = disp[("disp", s)].clone().requires_grad_(True)
            leaves.append(outputs[("disp", s)])
        for f in (-1, 1):
            aa, tr = gin.small_poses(rng, B)
            outputs[("cam_T_cam", 0, f)] = RL.transformation_from_parameters(aa, tr, invert=(f < 0)).detach()
        CP.Completor.generate_images_pred(ns, inp, outputs, opt.frame_ids)
        torch.manual_seed(1808)
        losses = CP.Completor.compute_losses(ns, inp, outputs)
        grads = torch.autograd.grad(losses["loss"], leaves)
        for k, v in losses.items():
            out[tag + "/L/" + k.replace("/", "_")] = np.array(float(v.detach()), dtype=np.float64)
        for s in range(4):
            put_grad(out, tag + "/g_disp%d" % s, grads[s])
        out[tag + "/siloss_weight_after"] = np.array(opt.completion_siloss_weight)
    # monitoring metrics: ground truth at the network resolution (KITTI completion crops), with and without the Garg crop
    gt, pred = gin.depth_eval_inputs(809, 2, H, W, gt_h=H, gt_w=W)
    for tag, crop in (("nocrop", False), ("crop", True)):
        ns = SimpleNamespace(opt=SimpleNamespace(completion_eigen_crop=crop), depth_metric_names=names)
        m = {}
        CP.Completor.compute_depth_losses(ns, {"depth_gt": torch.from_numpy(gt)}, {("depth", 0, 0): torch.from_numpy(pred)}, m)
        out["metrics_" + tag] = np.array([float(m[n]) for n in names], dtype=np.float64)
    save("completor_b1_352x1216", **out)


def gold_evaluate():
    """evaluate_depth.py:42-60 compute_errors and :62-70 batch_post_process_disparity, lifted out of the script by name (the
    module itself imports cv2 / matplotlib / the dataset stack and runs a whole evaluation at import-free call time only)."""
    src = open(os.path.join(REF, "evaluate_depth.py")).read()
    fns = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in ("compute_errors", "batch_post_process_disparity")]
    ns = {"np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "evaluate_depth.py", "exec"), ns)
    out = {"seed": np.array(4242)}
    for i, (gt, pred) in enumerate(gin.eval_pairs(4242)):
        out["errors%d" % i] = np.array(ns["compute_errors"](gt, pred), dtype=np.float64)
    l, r = gin.disp_pair(4243, 2, 192, 640)
    pp = ns["batch_post_process_disparity"](l, r)
    assert pp.dtype == np.float64
    out["pp_sub"] = pp[:, ::7, ::3].copy()
    out["pp_edges"] = np.concatenate([pp[:, :, :40], pp[:, :, -40:]], 2)[:, ::16].copy()
    out["pp_sum"] = np.array(pp.sum())
    save("evaluate_metrics", **out)


def gold_options():
    """Flag surface of the reference's argparse (options.py:9-480): name -> default/type/choices/action."""
    import json
    import options as ref_options
    parser = ref_options.MonodepthOptions().parser
    table = {}
    for a in parser._actions:
        if not a.option_strings or a.dest == "help":
            continue
        table[a.dest] = {"flag": a.option_strings[0], "default": a.default, "nargs": a.nargs,
                         "type": getattr(a.type, "__name__", None), "choices": list(a.choices) if a.choices else None,
                         "action": type(a).__name__}
    path = os.path.join(HERE, "options_surface.json")
    json.dump(table, open(path, "w"), indent=1, sort_keys=True)
    print("wrote options_surface.json (%d flags)" % len(table))


def main():
    torch.set_num_threads(8)
    RL, RT, DD, PD, PC, get2ch = load_reference()
    gold_options()
    gold_layers(RL)
    gold_decoders(DD, PD, PC)
    gold_losses(RL, RT, "losses_b2_64x96", 404, 2, 64, 96, full_arrays=True)
    gold_losses(RL, RT, "losses_b1_192x640", 505, 1, 192, 640, full_arrays=False)
    gold_losses(RL, RT, "losses_nossim_noautomask_b2_64x96", 606, 2, 64, 96, full_arrays=False,
                opt_over=dict(no_ssim=True, disable_automasking=True))
    gold_losses(RL, RT, "losses_emptysi_all_b2_64x96", 404, 2, 64, 96, full_arrays=False, empty_si="all")
    gold_losses(RL, RT, "losses_emptysi_scale2_b2_64x96", 404, 2, 64, 96, full_arrays=False, empty_si="scale2")
    gold_scatter(get2ch)
    gold_depth_losses(RL, RT)
    gold_refiner(RL, DD, PD)
    gold_rasterize()
    gold_completor(RL)
    gold_evaluate()


if __name__ == "__main__":
    main()
