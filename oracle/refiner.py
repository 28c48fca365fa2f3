"""Oracle restatement of the hot methods of reference ``refiner.py`` (fp32 CPU).  TEST INFRASTRUCTURE ONLY.

The refiner (BASELINE.json config 5) freezes the trained depth / pose networks and trains a second
decoder (``refine2d_decoder`` = DepthDecoder(road=True, catxy, deep)) that sees, at every scale, the coarse disparity
rescaled to the sparse LiDAR (median ratio inside the Garg-like crop), pseudo-3D coordinates (``Cat_xy``) and the max-pooled
2-channel LiDAR map; the loss is the trainer's photometric / smoothness loss plus a scale-invariant log loss against the
dense GDC depth ``inputs["inf_gdc"]``.

Follows refiner.py:299-382 (process_batch), :557-563 (siloss), :592-693 (compute_losses); generate_images_pred (:487-541) and
predict_poses (:383-448) are the trainer's (oracle.trainer).  Pinned to golden vectors produced by running the reference's
own methods (tests/golden/make_golden.py: gold_refiner), with the ResNet trunks supplied by oracle.networks (torchvision is
not installed - same caveat as for the trainer).  The tie-break noise is an explicit input, as in oracle.trainer.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import layers as L
from . import networks as N
from . import trainer as T


def default_opt(**over):
    """options.py defaults the refiner reads on top of the trainer's (options.py:262-330)."""
    o = T.default_opt(refine_iter=1, refine_iter_gama=0.8, refine_offset=False, refine_depthnet_with_beam="false",
                      catxy="true", refine2d_deep="true", refine_a0="true", gdc_loss_weight=0.008,
                      gdc_loss_only_on_scale_0=True, train_entire_net=False, refine_2d=True)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def build_models(opt, seed=0):
    """refiner.py:80-160: the trainer's six networks (frozen, eval mode) + the trainable refine2d_decoder."""
    m = T.build_models(opt, seed)
    m["refine2d_decoder"] = N.DepthDecoder(m["encoder"].num_ch_enc, opt.scales, road=True, catxy=(opt.catxy == "true"),
                                            deep=(opt.refine2d_deep == "true"))
    for k, net in m.items():
        net.train() if k == "refine2d_decoder" else net.eval()
    return m


def refine_inputs(opt, inputs, outputs):
    """refiner.py:316-348: per scale, the 6- (or 3-) channel map fed to the refine decoder:
    [coarse disparity rescaled to the LiDAR's metric scale, Cat_xy pseudo-3D coordinates, max-pooled 2-channel LiDAR]."""
    beam = inputs["4beam"]
    two_cha = inputs["2channel"]
    disp_0 = outputs[("disp", 0)]
    res = {}
    for scale in opt.scales:
        if opt.refine_a0 != "true":
            disp = outputs[("disp", scale)]
        else:
            disp = disp_0
            disp_0 = F.max_pool2d(disp_0, 2, ceil_mode=True)
        disp640 = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
        _, depth = L.disp_to_depth(disp640, opt.min_depth, opt.max_depth)
        mask = beam > 0
        crop = torch.zeros_like(mask)
        crop[:, :, 78:190, 23:617] = 1
        mask = mask * crop
        ratio = torch.median(beam[mask] * 100.0) / torch.median(depth[mask]).detach()
        depth = depth * ratio
        scaled_disp = (F.interpolate(1 / depth, disp.shape[2:], mode="bilinear", align_corners=False) - 0.01) / 9.9
        if scale != 0:
            two_cha = F.max_pool2d(two_cha, 2, ceil_mode=True)
        if opt.catxy == "true":
            for _ in range(scale):
                depth = F.max_pool2d(depth, 2, ceil_mode=True)
            xyz = L.cat_xy(depth, inputs[("inv_K", scale)])
            res[("disp", scale)] = torch.cat([scaled_disp, xyz, two_cha], 1)
        else:
            res[("disp", scale)] = torch.cat([scaled_disp, two_cha], 1)
    return res


def siloss(opt, pred, target):
    """refiner.py:557-563."""
    valid = ((target > 1e-3) * (pred < 80) * (pred > 1e-3) * ((pred - target).abs() < opt.gdc_loss_threshold)).detach()
    d = torch.log(pred[valid]) - torch.log(target[valid])
    return torch.sqrt((d ** 2).mean() - opt.si_var * (d.mean() ** 2)) * 10.0


def compute_losses(opt, inputs, outputs, losses, gama=1.0, noise=None):
    """refiner.py:592-693 (automask / min-reprojection default path)."""
    total = 0
    nscales = len(opt.scales)
    for s in opt.scales:
        src_s = s if opt.v1_multiscale else 0
        disp = outputs[("disp", s)].clone()
        color = inputs[("color", 0, s)]
        target = inputs[("color", 0, src_s)]
        reproj = torch.cat([T.reprojection_loss(opt, outputs[("color", f, s)], target) for f in opt.frame_ids[1:]], 1)
        if opt.avg_reprojection:
            reproj = reproj.mean(1, keepdim=True)
        loss = 0
        if not opt.disable_automasking:
            ident = torch.cat([T.reprojection_loss(opt, inputs[("color", f, src_s)], target) for f in opt.frame_ids[1:]], 1)
            if opt.avg_reprojection:
                ident = ident.mean(1, keepdim=True)
            eps = torch.randn(ident.shape) if noise is None else noise[s]
            ident = ident + eps * 0.00001
            combined = torch.cat((ident, reproj), dim=1)
        else:
            combined = reproj
        if combined.shape[1] == 1:
            to_opt = combined
        else:
            to_opt, idxs = torch.min(combined, dim=1)
        if not opt.disable_automasking:
            outputs["identity_selection/{}".format(s)] = (idxs > ident.shape[1] - 1).float()
        loss = loss + to_opt.mean()
        mean_disp = disp.mean(2, True).mean(3, True)
        norm_disp = disp / (mean_disp + 1e-7)
        loss = loss + opt.disparity_smoothness * L.get_smooth_loss(norm_disp, color) / (2 ** s)
        total = total + loss
        losses["loss/gama{}_scale{}".format(gama, s)] = loss
        if (not opt.gdc_loss_only_on_scale_0) or s == 0:
            gdc = inputs["inf_gdc"].squeeze()
            d = F.interpolate(disp, [192, 640], mode="bilinear", align_corners=False).squeeze()
            _, depth = L.disp_to_depth(d, opt.min_depth, opt.max_depth)
            gdc_loss = siloss(opt, depth, gdc) * opt.gdc_loss_weight
            if opt.gdc_loss_only_on_scale_0:
                gdc_loss = gdc_loss * 4.0
            total = total + gdc_loss
            losses["loss/gdc_scale{}".format(s)] = gdc_loss
    total = total / nscales
    losses["loss"] = losses["loss"] + total * gama
    return losses


def process_batch(opt, models, inputs, noise=None):
    """refiner.py:299-382 (train_entire_net=False, refine_2d, separate_resnet pose net, beam_encoder)."""
    with torch.no_grad():
        features = models["encoder"](inputs[("color_aug", 0, 0)])
        beam_features = models["beam_encoder"](inputs["2channel"])
        if opt.refine_depthnet_with_beam == "true":
            outputs = dict(models["depth"](features, beam_features=beam_features))
        else:
            outputs = dict(models["depth"](features))
    outputs.update(refine_inputs(opt, inputs, outputs))
    outputs.update(T.predict_poses(opt, models, inputs))
    losses = {"loss": 0.0}
    n_iter = opt.refine_iter
    for it in range(n_iter):
        offset = models["refine2d_decoder"](features, beam_features=beam_features, depth_maps=outputs, tanh=opt.refine_offset)
        for s in opt.scales:
            outputs[("disp", s)] = offset[("disp", s)]
        T.generate_images_pred(opt, inputs, outputs)
        gama_base = 1.0 if n_iter == 1 else opt.refine_iter_gama
        losses = compute_losses(opt, inputs, outputs, losses, gama=gama_base ** (n_iter - it), noise=noise)
    return outputs, losses
