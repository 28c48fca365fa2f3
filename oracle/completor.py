"""Oracle restatement of the hot methods of reference ``completor.py`` (fp32 CPU).  TEST INFRASTRUCTURE ONLY.

The completion driver runs the trainer's graph (completor.py:277-388 process_batch / predict_poses and :428-476
generate_images_pred are line-for-line the trainer's, restated in ``oracle.trainer``) with
  * its own hyper-parameters (completor.py:31-34, 58-89, 120-122),
  * its LiDAR term (completor.py:621-725): scale 0 only unless ``completion_siloss_all_scale == "true"``; SI-log loss, or
    a masked L1 * 0.001 when ``completion_siloss`` is off and ``completion_l1loss`` on,
  * its monitoring metrics (completor.py:728-762).
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import layers as L
from . import networks as N
from . import trainer as OT


def default_opt(**over):
    """options.py:333-391 completion defaults + the trainer options the shared graph reads."""
    o = vars(OT.default_opt())
    o.update(height=352, width=1216, completion_num_layers=50, completion_pose_num_layers=18, completion_siloss=True,
             completion_l1loss=False, completion_siloss_all_scale="false", completion_siloss_weight=0.1,
             completion_eigen_crop=False, completion_scheduler_step_size=25, completion_num_epochs=3,
             completion_not_full_res=False, learning_rate=1e-4, batch_size=12)
    o.update(over)
    return SimpleNamespace(**o)


def build_models(opt, seed=0):
    """completor.py:58-104 (default flags: separate_resnet pose net, beam_encoder)."""
    torch.manual_seed(seed)
    m = {}
    m["encoder"] = N.ResnetEncoder(opt.completion_num_layers, False)
    m["beam_encoder"] = N.ResnetEncoder(opt.completion_num_layers, False, beam_encoder=True)
    m["beam_encoder_pose"] = N.ResnetEncoder(opt.completion_pose_num_layers, False, num_input_images=2, beam_encoder=True)
    m["depth"] = N.DepthDecoder(m["encoder"].num_ch_enc, opt.scales)
    m["pose_encoder"] = N.ResnetEncoder(opt.completion_pose_num_layers, False, num_input_images=2)
    m["pose"] = N.PoseDecoder(m["pose_encoder"].num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
    return m


def lidar_term(opt, disp, beam):
    """completor.py:621-725 for one scale -> (name, value) or None."""
    disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
    _, depth = L.disp_to_depth(disp, opt.min_depth, opt.max_depth)
    beam_depth = beam * 100.0
    depth = depth * 26.0
    if opt.completion_siloss:
        mask = ((beam_depth > 1) & (depth < 80) & (depth > 1) & ((depth - beam_depth).abs() < opt.gdc_loss_threshold)).detach()
        d = torch.log(depth[mask]) - torch.log(beam_depth[mask])
        return "si_loss", torch.sqrt((d ** 2).mean() - opt.si_var * (d.mean() ** 2)) * 0.1
    if opt.completion_l1loss:
        mask = ((beam_depth > 1) & (depth < 80) & (depth > 1)).detach()
        return "l1_loss", (depth[mask] - beam_depth[mask]).abs().mean() * 0.001
    return None


def compute_losses(opt, inputs, outputs, noise=None):
    """completor.py:546-726: the trainer's photometric + smoothness terms, then this driver's LiDAR term."""
    t_opt = SimpleNamespace(**vars(opt))
    t_opt.trainer_siloss = "false"
    losses = OT.compute_losses(t_opt, inputs, outputs, noise)
    total = losses["loss"] * len(opt.scales)
    for s in opt.scales:
        if opt.completion_siloss_all_scale == "true" or s == 0:
            term = lidar_term(opt, outputs[("disp", s)], inputs["4beam"])
            if term is not None:
                total = total + term[1]
                losses["loss/{}{}".format(term[0], s)] = term[1]
    losses["loss"] = total / len(opt.scales)
    return losses


def process_batch(opt, models, inputs, noise=None):
    """completor.py:277-318 (default flags)."""
    feats = models["encoder"](inputs[("color_aug", 0, 0)])
    outputs = dict(models["depth"](feats, beam_features=models["beam_encoder"](inputs["2channel"])))
    outputs.update(OT.predict_poses(opt, models, inputs))
    OT.generate_images_pred(opt, inputs, outputs)
    return outputs, compute_losses(opt, inputs, outputs, noise)


def compute_depth_losses(opt, depth_pred, depth_gt):
    """completor.py:728-762 -> the 7 metrics, errors on millimetres."""
    gh, gw = depth_gt.shape[2:]
    pred = torch.clamp(F.interpolate(depth_pred, [gh, gw], mode="bilinear", align_corners=False), 1e-3, 80).detach()
    mask = depth_gt > 0.1
    if opt.completion_eigen_crop:
        crop = torch.zeros_like(mask)
        crop[:, :, 153:371, 44:1197] = 1
        mask = mask * crop
    gt, pr = depth_gt[mask], pred[mask]
    pr = torch.clamp(pr * (torch.median(gt) / torch.median(pr)), min=1e-3, max=80)
    return [float(v) for v in L.compute_depth_errors(gt * 1000.0, pr * 1000.0)]
