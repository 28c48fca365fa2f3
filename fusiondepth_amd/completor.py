"""MI355X-native ``Completor`` — the reference's depth-completion driver (``completor.py``) on the trainer's kernels.

The completion step is the trainer's graph at KITTI-completion resolution (1216x352 unless ``--completion_not_full_res``)
with its own hyper-parameters and LiDAR term; what differs from ``Trainer`` (reference lines cited per member):

  * encoders: ``--completion_num_layers`` (default ResNet-50) for the colour / beam encoders and
    ``--completion_pose_num_layers`` for the pose encoders (completor.py:58-89);
  * optimiser: plain ``--learning_rate``, StepLR(``--completion_scheduler_step_size``), ``--completion_num_epochs`` and one
    optimiser step per batch - no gradient accumulation, no batch-size-derived rescaling (completor.py:120-122, 229-246);
  * LiDAR term (completor.py:621-725): evaluated at scale 0 only unless ``--completion_siloss_all_scale true``; the SI-log
    loss by default, or with ``--completion_siloss`` (a store_false flag) and ``--completion_l1loss`` a masked L1 * 0.001
    without the |pred - beam| gate (``fd_photo_cfg.si_mode = 1``);
  * monitoring metrics (completor.py:728-762): ground truth valid where > 0.1, Garg crop only with
    ``--completion_eigen_crop``, errors computed on millimetres (x 1000);
  * validation (completor.py:390-426): mean metrics over the validation batches, best-RMSE bookkeeping that names the
    checkpoint ``rms<N>``.
The wandb / DataLoader shell (completor.py:131-146, 767-790) is out of scope like the trainer's.
"""
import numpy as np
import torch

from . import functional as FD
from .trainer import Trainer


class Completor(Trainer):
    def __init__(self, options, device=None, rank=0, world_size=1, materialize_outputs=False, verbose=True):
        if not options.completion_not_full_res:                          # completor.py:31-34
            options.height, options.width = 352, 1216
        super().__init__(options, device=device, rank=rank, world_size=world_size, materialize_outputs=materialize_outputs,
                         verbose=verbose)
        self.best = 100000.0                                             # completor.py:186

    # ---- configuration hooks of Trainer.__init__ ----------------------------------------------------
    def _derived_hparams(self, vram_gib):
        o = self.opt
        return dict(num_epochs=o.completion_num_epochs, accumulate_step=1, learning_rate=o.learning_rate,
                    scheduler_step_size=o.completion_scheduler_step_size, micro_batch=o.batch_size)

    def _encoder_layers(self):
        return self.opt.completion_num_layers, self.opt.completion_pose_num_layers

    def _lidar_term(self):
        o = self.opt
        scales = list(o.scales) if o.completion_siloss_all_scale == "true" else [s for s in o.scales if s == 0]
        if o.completion_siloss:
            return scales, 0, "si_loss"
        if o.completion_l1loss:
            return scales, 1, "l1_loss"
        return [], 0, "si_loss"

    def compute_losses(self, inputs, outputs):
        """completor.py:546-726.  (completor.py:694-695 also doubles ``opt.completion_siloss_weight`` on every call when the
        term is scale-0 only; that option feeds no arithmetic, the side effect is kept for option-dump parity.)"""
        if self.opt.completion_siloss_all_scale != "true":
            self.opt.completion_siloss_weight *= 2.0
        return super().compute_losses(inputs, outputs)

    # ---- monitoring ----------------------------------------------------------------------------------
    def compute_depth_losses(self, inputs, outputs, losses, accumulate=False):
        """completor.py:728-762."""
        depth_pred = outputs[("depth", 0, 0)].detach()
        depth_gt = inputs["depth_gt"]
        gt_h, gt_w = depth_gt.shape[2:]
        if (gt_h, gt_w) != tuple(depth_pred.shape[2:]):
            depth_pred = FD.bilinear_upsample(depth_pred, (gt_h, gt_w)) if gt_h >= depth_pred.shape[2] else \
                torch.nn.functional.interpolate(depth_pred, [gt_h, gt_w], mode="bilinear", align_corners=False)
        depth_pred = torch.clamp(depth_pred, 1e-3, 80)
        mask = depth_gt > 0.1
        if self.opt.completion_eigen_crop:
            crop = torch.zeros_like(mask)
            crop[:, :, 153:371, 44:1197] = 1
            mask = mask * crop
        gt, pred = depth_gt[mask], depth_pred[mask]
        pred = torch.clamp(pred * (torch.median(gt) / torch.median(pred)), min=1e-3, max=80)
        errs = FD.depth_errors(gt * 1000.0, pred * 1000.0)
        for i, metric in enumerate(self.depth_metric_names):
            v = errs[i].detach().cpu().numpy()
            losses[metric] = losses.get(metric, 0.0) + v if accumulate else v

    def val(self, batches, save_best=True):
        """completor.py:390-426: mean metrics over ``batches``; a new best de/rms is remembered and, below 1200 (mm), saved
        as ``weights_rms<N>``.  Returns (losses, checkpoint folder or None)."""
        losses = self.val_metrics(batches)
        saved = None
        if losses["de/rms"] < self.best:
            self.best = float(losses["de/rms"])
            rms = round(float(losses["de/rms"]))
            if save_best and rms < 1200:
                saved = self.save_model("rms{}".format(rms))
        self.last_saved = [saved] if saved else []
        return losses, saved
