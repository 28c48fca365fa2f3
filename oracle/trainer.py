"""Oracle restatement of the hot methods of reference ``trainer.py`` (fp32 CPU).

Follows trainer.py:268-319 (process_batch, incl. the ``shared`` branch :275-283 and the predictive-mask decoder
:305-306), :321-388 (predict_poses: "pairs" / "all" inputs, "separate_resnet" / "shared" networks, the stereo partner "s"
skipped), :425-474 (generate_images_pred, ``stereo_T`` for "s"), :476-488
(compute_reprojection_loss), :490-596 (compute_losses incl. the predictive mask :530-541) and the step in
:237-248.  The tie-break noise (trainer.py:549-552) is an explicit *input*
(``noise[scale]`` of shape [B,2,H,W]) so runs are reproducible; pass ``None`` to
draw it from torch's global generator exactly like the reference does.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import layers as L
from . import networks as N


def default_opt(**over):
    """Subset of options.py defaults the hot path reads (options.py:34-79,239-330)."""
    o = dict(height=192, width=640, scales=[0, 1, 2, 3], frame_ids=[0, -1, 1], min_depth=0.1, max_depth=100.0,
             disparity_smoothness=1e-3, no_ssim=False, disable_automasking=False, avg_reprojection=False,
             v1_multiscale=False, trainer_siloss="true", trainer_siloss_all_scale=True, gdc_loss_threshold=2.0,
             si_var=0.3, beam_encoder=True, num_layers=18, learning_rate=1e-4, batch_size=12,
             scheduler_step_size=10, use_stereo=False, predictive_mask=False, pose_model_type="separate_resnet",
             pose_model_input="pairs")
    o.update(over)
    return SimpleNamespace(**o)


def derived_hparams(batch_size, learning_rate=1e-4, scheduler_step_size=10, vram_gib=288.0):
    """trainer.py:28-41 — epochs, accumulate_step, lr, StepLR step, micro-batch."""
    accumulate = 2 if vram_gib < 15 else 1
    if batch_size > 8:
        accumulate *= 2
    return SimpleNamespace(num_epochs=(8 * 17) // batch_size, accumulate_step=accumulate,
                           learning_rate=learning_rate * (batch_size / 8),
                           scheduler_step_size=int(scheduler_step_size * (8 / batch_size)),
                           micro_batch=int(batch_size / accumulate))


def frame_setup(opt):
    """trainer.py:56-64: (num_pose_frames, use_pose_net, frame ids incl. the stereo partner "s")."""
    fids = [f for f in opt.frame_ids if f != "s"]
    num_pose_frames = 2 if getattr(opt, "pose_model_input", "pairs") == "pairs" else len(fids)
    stereo = getattr(opt, "use_stereo", False)
    use_pose_net = not (stereo and fids == [0])
    return num_pose_frames, use_pose_net, fids + (["s"] if stereo else [])


def build_models(opt, seed=0):
    """trainer.py:66-127 — the networks in the reference's construction order (weights_init=scratch)."""
    torch.manual_seed(seed)
    npf, use_pose_net, frame_ids = frame_setup(opt)
    kind = getattr(opt, "pose_model_type", "separate_resnet")
    m = {}
    m["encoder"] = N.ResnetEncoder(opt.num_layers, False)
    if opt.beam_encoder:
        m["beam_encoder"] = N.ResnetEncoder(opt.num_layers, False, beam_encoder=True)
        m["beam_encoder_pose"] = N.ResnetEncoder(opt.num_layers, False, num_input_images=npf, beam_encoder=True)
    m["depth"] = N.DepthDecoder(m["encoder"].num_ch_enc, opt.scales)
    if use_pose_net and kind == "separate_resnet":
        m["pose_encoder"] = N.ResnetEncoder(opt.num_layers, False, num_input_images=npf)
        m["pose"] = N.PoseDecoder(m["pose_encoder"].num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
    elif use_pose_net and kind == "shared":
        m["pose"] = N.PoseDecoder(m["encoder"].num_ch_enc, npf)                       # trainer.py:106-108
    if getattr(opt, "predictive_mask", False):                                        # trainer.py:117-127
        m["predictive_mask"] = N.DepthDecoder(m["encoder"].num_ch_enc, opt.scales, num_output_channels=len(frame_ids) - 1)
    return m


def predict_poses(opt, models, inputs, features=None):
    """trainer.py:321-388.  ``features``: frame id -> depth-encoder features, for the ``shared`` pose network."""
    out = {}
    npf, _, frame_ids = frame_setup(opt)
    kind = getattr(opt, "pose_model_type", "separate_resnet")
    if npf == 2:
        for f in frame_ids[1:]:
            if f == "s":
                continue
            order = (f, 0) if f < 0 else (0, f)          # always temporal order
            if kind == "shared":
                axisangle, translation = models["pose"]([features[i] for i in order])
            else:
                rgb = torch.cat([inputs[("color_aug", i, 0)] for i in order], 1)
                pose_in = [models["pose_encoder"](rgb)]
                if opt.beam_encoder:
                    lidar = torch.cat([inputs[("2channel", i, 0)] for i in order], 1)
                    beam_in = [models["beam_encoder_pose"](lidar)]
                    axisangle, translation = models["pose"](pose_in, beam_inputs=beam_in)
                else:
                    axisangle, translation = models["pose"](pose_in)
            out[("axisangle", 0, f)] = axisangle
            out[("translation", 0, f)] = translation
            out[("cam_T_cam", 0, f)] = L.transformation_from_parameters(axisangle[:, 0], translation[:, 0],
                                                                        invert=(f < 0))
    else:
        if kind == "shared":
            pose_in = [features[i] for i in frame_ids if i != "s"]
        else:
            pose_in = [models["pose_encoder"](torch.cat([inputs[("color_aug", i, 0)] for i in frame_ids if i != "s"], 1))]
        axisangle, translation = models["pose"](pose_in)
        for i, f in enumerate(frame_ids[1:]):
            if f != "s":
                out[("axisangle", 0, f)] = axisangle
                out[("translation", 0, f)] = translation
                out[("cam_T_cam", 0, f)] = L.transformation_from_parameters(axisangle[:, i], translation[:, i])
    return out


def generate_images_pred(opt, inputs, outputs):
    """trainer.py:425-474 (non-stereo, non-posecnn path)."""
    for s in opt.scales:
        disp = outputs[("disp", s)]
        if opt.v1_multiscale:
            src_s = s
        else:
            disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
            src_s = 0
        _, depth = L.disp_to_depth(disp, opt.min_depth, opt.max_depth)
        outputs[("depth", 0, s)] = depth
        h, w = depth.shape[2:]
        for f in frame_setup(opt)[2][1:]:
            T = inputs["stereo_T"] if f == "s" else outputs[("cam_T_cam", 0, f)]
            pts = L.backproject_depth(depth, inputs[("inv_K", src_s)])
            grid = L.project_3d(pts, inputs[("K", src_s)], T, h, w)
            outputs[("sample", f, s)] = grid
            outputs[("color", f, s)] = F.grid_sample(inputs[("color", f, src_s)], grid, padding_mode="border",
                                                     align_corners=False)
            if not opt.disable_automasking:
                outputs[("color_identity", f, s)] = inputs[("color", f, src_s)]


def reprojection_loss(opt, pred, target):
    """trainer.py:476-488."""
    l1 = (target - pred).abs().mean(1, True)
    if opt.no_ssim:
        return l1
    return 0.85 * L.ssim(pred, target).mean(1, True) + 0.15 * l1


def si_log_loss(opt, disp, beam):
    """trainer.py:577-589 — masked scale-invariant log loss vs the 4-beam LiDAR."""
    disp = F.interpolate(disp, [opt.height, opt.width], mode="bilinear", align_corners=False)
    _, depth = L.disp_to_depth(disp, opt.min_depth, opt.max_depth)
    beam_depth = beam * 100.0
    depth = depth * 26.0
    mask = ((beam_depth > 1) & (depth < 80) & (depth > 1) & ((depth - beam_depth).abs() < opt.gdc_loss_threshold)).detach()
    d = torch.log(depth[mask]) - torch.log(beam_depth[mask])
    return torch.sqrt((d ** 2).mean() - opt.si_var * (d.mean() ** 2)) * 0.1


def compute_losses(opt, inputs, outputs, noise=None):
    """trainer.py:490-596 (automask / min-reprojection default path)."""
    losses = {}
    total = 0
    frame_ids = frame_setup(opt)[2]
    for s in opt.scales:
        src_s = s if opt.v1_multiscale else 0
        disp = outputs[("disp", s)]
        color = inputs[("color", 0, s)]
        target = inputs[("color", 0, src_s)]
        reproj = torch.cat([reprojection_loss(opt, outputs[("color", f, s)], target) for f in frame_ids[1:]], 1)
        loss = 0
        if opt.disable_automasking and getattr(opt, "predictive_mask", False):        # trainer.py:530-541
            mask = outputs["predictive_mask"][("disp", s)]
            if not opt.v1_multiscale:
                mask = F.interpolate(mask, [opt.height, opt.width], mode="bilinear", align_corners=False)
            reproj = reproj * mask
            loss = loss + 0.2 * F.binary_cross_entropy(mask, torch.ones_like(mask))
        if opt.avg_reprojection:
            reproj = reproj.mean(1, keepdim=True)
        if not opt.disable_automasking:
            ident = torch.cat([reprojection_loss(opt, inputs[("color", f, src_s)], target)
                               for f in frame_ids[1:]], 1)
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
        losses["loss/{}".format(s)] = loss
        if opt.trainer_siloss == "true" and (opt.trainer_siloss_all_scale or s == 0):
            si = si_log_loss(opt, disp, inputs["4beam"])
            total = total + si
            losses["loss/si_loss{}".format(s)] = si
    losses["loss"] = total / len(opt.scales)
    return losses


def process_batch(opt, models, inputs, noise=None):
    """trainer.py:268-319."""
    _, use_pose_net, frame_ids = frame_setup(opt)
    if getattr(opt, "pose_model_type", "separate_resnet") == "shared":                # trainer.py:275-283
        B = inputs[("color_aug", 0, 0)].shape[0]
        all_feats = models["encoder"](torch.cat([inputs[("color_aug", i, 0)] for i in frame_ids]))
        feats = {k: [f[i * B:(i + 1) * B] for f in all_feats] for i, k in enumerate(frame_ids)}
        outputs = dict(models["depth"](feats[0]))
    else:
        feats = models["encoder"](inputs[("color_aug", 0, 0)])
        if opt.beam_encoder:
            outputs = models["depth"](feats, beam_features=models["beam_encoder"](inputs["2channel"]))
        else:
            outputs = models["depth"](feats)
        outputs = dict(outputs)
    if getattr(opt, "predictive_mask", False):                                        # trainer.py:305-306
        outputs["predictive_mask"] = dict(models["predictive_mask"](feats))
    if use_pose_net:
        outputs.update(predict_poses(opt, models, inputs, feats))
    generate_images_pred(opt, inputs, outputs)
    return outputs, compute_losses(opt, inputs, outputs, noise)


def compute_depth_losses(depth_pred, depth_gt):
    """Reference: trainer.py:598-630 — monitoring metrics of the scale-0 depth against sparse ground truth: bilinear resize
    to the ground-truth size, clamp to [1e-3, 80], Garg/Eigen crop rows 153..370 / cols 44..1196 of valid (> 0) pixels,
    median scaling, clamp again, layers.compute_depth_errors.  Returns the 7 metrics (abs_rel, sq_rel, rms, log_rms, a1-3)."""
    import torch.nn.functional as F
    from . import layers as OL
    gh, gw = depth_gt.shape[2:]
    pred = torch.clamp(F.interpolate(depth_pred, [gh, gw], mode="bilinear", align_corners=False), 1e-3, 80).detach()
    mask = depth_gt > 0
    crop = torch.zeros_like(mask)
    crop[:, :, 153:371, 44:1197] = 1
    mask = mask * crop
    gt, pr = depth_gt[mask], pred[mask]
    pr = torch.clamp(pr * (torch.median(gt) / torch.median(pr)), min=1e-3, max=80)
    return [float(v) for v in OL.compute_depth_errors(gt, pr)]


def trainable_parameters(models):
    """trainer.py:71-127 order: encoder, beam_encoder, beam_encoder_pose, depth, pose_encoder, pose."""
    params = []
    for k in ["encoder", "beam_encoder", "beam_encoder_pose", "depth", "pose_encoder", "pose", "predictive_mask"]:
        if k in models:
            params += list(models[k].parameters())
    return params


class OracleTrainer:
    """Minimal harness = trainer.py:129-131 (Adam + StepLR) + :237-248 (accumulate/step)."""

    def __init__(self, opt, seed=0, models=None):
        self.opt = opt
        self.hp = derived_hparams(opt.batch_size, opt.learning_rate, opt.scheduler_step_size)
        self.models = models if models is not None else build_models(opt, seed)
        for m in self.models.values():
            m.train()
        self.optimizer = torch.optim.Adam(trainable_parameters(self.models), self.hp.learning_rate)
        self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, max(self.hp.scheduler_step_size, 1), 0.1)
        self.optimizer.zero_grad()
        self.batch_idx = 0

    def micro_step(self, inputs, noise=None):
        outputs, losses = process_batch(self.opt, self.models, inputs, noise)
        loss = losses["loss"] / self.hp.accumulate_step
        loss.backward()
        if (self.batch_idx + 1) % self.hp.accumulate_step == 0:
            self.optimizer.step()
            self.optimizer.zero_grad()
        self.batch_idx += 1
        return outputs, losses
