"""MI355X-native ``Trainer`` — same class/method surface as the reference ``trainer.py`` (Trainer.__init__,
set_train/set_eval, process_batch, predict_poses, generate_images_pred, compute_reprojection_loss,
compute_losses, compute_depth_losses, save_model/load_model), driven by ``options.MonodepthOptions``.

What differs, by design:
  * every op is a libfdhip kernel; per pyramid scale, generate_images_pred + the photometric/SI part of
    compute_losses are ONE fused kernel (csrc/photometric.hip) — the reference's ("depth",0,s), ("sample",f,s),
    ("color",f,s) tensors are only materialised when ``materialize_outputs`` is set (logging / tests);
  * all trainable tensors live in one flat buffer: Adam is one fused launch, and data-parallel training
    (one process per GPU, RCCL all-reduce over xGMI, see dp.py) is added — the reference is single-GPU;
  * the data loader / wandb / tensorboard shell of the reference is out of scope (SURVEY.md §2 rows 11,12):
    batches are dicts with the reference's schema, e.g. from ``synthetic.make_batch``.
Reference lines are cited per method.
"""
import json
import contextlib
import os
import time

import numpy as np
import torch

from . import dp
from . import functional as FD
from . import networks
from . import tuning
from .layers import SSIM, BackprojectDepth, Project3D, disp_to_depth, transformation_from_parameters

MODEL_ORDER = ["encoder", "beam_encoder", "beam_encoder_pose", "depth", "pose_encoder", "pose", "predictive_mask"]


def _hms(t):
    """utils.py:37-42."""
    t = int(t)
    return "{:02d}h{:02d}m{:02d}s".format(t // 3600, (t % 3600) // 60, t % 60)


class Outputs(dict):
    """The reference's ``outputs`` dict.  Entries the fused loss kernel does not need to materialise are derived on first
    access instead of costing launches every step: ``"identity_selection/<s>"`` (trainer.py:564-565, a monitoring image) from
    the kernel's argmin index, ``("depth", 0, s)`` (trainer.py:430-438, read by compute_depth_losses) from ``("disp", s)``."""

    def __init__(self, *args, depth_spec=None, **kw):
        super().__init__(*args, **kw)
        self.depth_spec = depth_spec       # (height, width, min_depth, max_depth, v1_multiscale)

    @classmethod
    def for_options(cls, opt, *args):
        return cls(*args, depth_spec=(opt.height, opt.width, opt.min_depth, opt.max_depth, bool(getattr(opt, "v1_multiscale", False))))

    @staticmethod
    def _selection_scale(key):
        """"identity_selection/<s>" -> s, or None for any other key."""
        if isinstance(key, str) and key.startswith("identity_selection/"):
            tail = key.split("/", 1)[1]
            if tail.isdigit():
                return int(tail)
        return None

    def __missing__(self, key):
        if isinstance(key, tuple) and len(key) == 3 and key[0] == "depth" and key[1] == 0 and self.depth_spec is not None:
            disp = dict.get(self, ("disp", key[2]))
            if disp is not None:
                H, W, lo, hi, v1 = self.depth_spec
                with torch.no_grad():
                    up = disp.detach()
                    if not v1 and tuple(up.shape[2:]) != (H, W):       # --v1_multiscale keeps the scale's own resolution
                        up = FD.bilinear_upsample(up, (H, W))
                    val = disp_to_depth(up, lo, hi)[1]
                self[key] = val
                return val
        s = self._selection_scale(key)
        if s is not None:
            raw = dict.get(self, ("sel", s))
            if raw is not None:
                sel, n_id = raw
                val = (sel > n_id - 1).float()
                self[key] = val
                return val
        raise KeyError(key)

    def __contains__(self, key):
        if dict.__contains__(self, key):
            return True
        s = self._selection_scale(key)
        return s is not None and dict.__contains__(self, ("sel", s))


def derived_hparams(opt, vram_gib):
    """trainer.py:28-41: epochs, accumulate_step, lr, StepLR step and micro-batch derived from --batch_size."""
    accumulate = 2 if vram_gib < 15 else 1
    if opt.batch_size > 8:
        accumulate *= 2
    return dict(num_epochs=(8 * 17) // opt.batch_size, accumulate_step=accumulate,
                learning_rate=opt.learning_rate * (opt.batch_size / 8),
                scheduler_step_size=int(opt.scheduler_step_size * (8 / opt.batch_size)),
                micro_batch=int(opt.batch_size / accumulate))


class Trainer:
    _pose_on_main = False      # True only inside train_step_graphed: the captured step keeps every fork on the capture stream

    def __init__(self, options, device=None, rank=0, world_size=1, materialize_outputs=False, verbose=True):
        self.opt = options
        if self.opt.no_cuda or not torch.cuda.is_available():
            raise RuntimeError("fusiondepth_amd.Trainer needs an MI355X: there is no CPU path (use oracle/ for CPU checks)")
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is not None:
            torch.cuda.set_device(self.device)           # libfdhip launches on the CURRENT device's streams (see _lib.stream)
        self.verbose = verbose
        self.rank, self.world_size = rank, world_size
        self.materialize_outputs = materialize_outputs

        vram = torch.cuda.get_device_properties(self.device).total_memory / 1024 ** 3      # trainer.py:30-35
        hp = self._derived_hparams(vram)
        self.opt.num_epochs = hp["num_epochs"]                                                # trainer.py:28
        self.accumulate_step = hp["accumulate_step"]
        self.learning_rate = hp["learning_rate"]
        self.scheduler_step_size = hp["scheduler_step_size"]
        self.batch_size = hp["micro_batch"]                                                   # per-process micro-batch
        self.log_path = os.path.join(self.opt.log_dir, self.opt.model_name)

        assert self.opt.height % 32 == 0, "'height' must be a multiple of 32"                 # trainer.py:47-48
        assert self.opt.width % 32 == 0, "'width' must be a multiple of 32"
        assert self.opt.frame_ids[0] == 0, "frame_ids must start with 0"
        self.num_scales = len(self.opt.scales)
        self.num_input_frames = len(self.opt.frame_ids)
        self.num_pose_frames = 2 if self.opt.pose_model_input == "pairs" else self.num_input_frames
        self.use_pose_net = not (self.opt.use_stereo and self.opt.frame_ids == [0])          # trainer.py:61
        if self.opt.use_stereo:
            self.opt.frame_ids = list(self.opt.frame_ids) + ["s"]                            # trainer.py:63-64
        shared = self.opt.pose_model_type == "shared"
        if shared and self.use_pose_net and self.opt.beam_encoder and self.num_pose_frames == 2:
            raise NotImplementedError(
                "--pose_model_type shared with --pose_model_input pairs and the LiDAR encoders: the reference reads "
                "`beam_pose_feats`, which only its non-shared branch defines (trainer.py:330-346), i.e. it stops with a NameError "
                "- there is no behaviour to reproduce; pass --beam_encoder false")
        if shared and self.opt.predictive_mask:
            raise NotImplementedError("--pose_model_type shared with --predictive_mask: the reference hands the per-frame feature "
                                      "DICT to the mask decoder (trainer.py:275-283, 305-306) and fails there")
        if shared and (self.opt.cat_4beam_to_color or self.opt.cat2start):
            raise NotImplementedError("--pose_model_type shared feeds color_aug alone to the depth encoder (trainer.py:275-283); "
                                      "it cannot be combined with --cat_4beam_to_color / --cat2start (their stems have 4 / 5 inputs)")
        if self.opt.predictive_mask:
            assert self.opt.disable_automasking, \
                "When using predictive_mask, please disable automasking with --disable_automasking"     # trainer.py:118-119

        # ---- networks (trainer.py:66-127) --------------------------------------------------------------
        pre = self.opt.weights_init == "pretrained"
        depth_layers, pose_layers = self._encoder_layers()
        m = {}
        m["encoder"] = networks.ResnetEncoder(depth_layers, pre, cat4beam_to_color=self.opt.cat_4beam_to_color,
                                              cat2channel=self.opt.cat2start)
        if self.opt.beam_encoder:
            m["beam_encoder"] = networks.ResnetEncoder(depth_layers, pre, beam_encoder=True)
            m["beam_encoder_pose"] = networks.ResnetEncoder(pose_layers, pre, num_input_images=self.num_pose_frames,
                                                            beam_encoder=True)
        m["depth"] = networks.DepthDecoder(m["encoder"].num_ch_enc, self.opt.scales, cat2end=self.opt.cat2end)
        if self.use_pose_net and self.opt.pose_model_type == "separate_resnet":
            m["pose_encoder"] = networks.ResnetEncoder(pose_layers, pre, num_input_images=self.num_pose_frames)
            m["pose"] = networks.PoseDecoder(m["pose_encoder"].num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
        elif self.use_pose_net and self.opt.pose_model_type == "shared":                      # trainer.py:106-108
            m["pose"] = networks.PoseDecoder(m["encoder"].num_ch_enc, self.num_pose_frames)
        elif self.use_pose_net and self.opt.pose_model_type == "posecnn":
            m["pose"] = networks.PoseCNN(self.num_input_frames if self.opt.pose_model_input == "all" else 2)
        if self.opt.predictive_mask:                                                          # trainer.py:117-127
            m["predictive_mask"] = networks.DepthDecoder(m["encoder"].num_ch_enc, self.opt.scales,
                                                         num_output_channels=len(self.opt.frame_ids) - 1)
        for k in ("pose_encoder", "beam_encoder_pose"):           # PoseDecoder reads features[-1] only: their features[0] has no reader
            if k in m:
                m[k].stem_feature_needed = False
        self.models = {k: m[k].to(self.device) for k in MODEL_ORDER if k in m}
        if world_size > 1:
            dp.broadcast_module_state(self.models.values())
        self.parameters_to_train = []
        for k in self.models:
            self.parameters_to_train += list(self.models[k].parameters())

        # ---- optimiser (trainer.py:129-131): Adam + StepLR(gamma 0.1), on one flat buffer ----------------
        self.flat = dp.FlatParameters(self.parameters_to_train)
        FD.evict_dead_weight_layouts()         # cached layouts / re-layout plan of trainers that no longer exist
        FD.enable_weight_cache(self.parameters_to_train)
        FD.enable_direct_grad(self.parameters_to_train)
        # the depth decoder is the serial section of the step: its weight gradients leave the main stream (functional.enable_side_wgrad)
        for k in tuning.host.side_wgrad:
            if k in self.models:
                FD.enable_side_wgrad(self.models[k].parameters())
        self.exp_avg = torch.zeros_like(self.flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat_param)
        self.adam_step_count = 0
        self.lr = self.learning_rate
        self.adam_state = torch.tensor([0.0, self.lr], device=self.device)      # [step, lr] on the device (graph-safe)
        self._graph = None
        self._streams = []
        self.parallel_streams = True
        # the accumulated micro-batches run as one stacked pass where the step has the default shape (see train_step);
        # the flag variants without a separate pose encoder per frame pair run them one after the other like the reference
        self.stack_microbatches = (self.use_pose_net and self.opt.pose_model_type == "separate_resnet"
                                   and self.num_pose_frames == 2)
        # opt-in (tuning.host.interleave): the four encoders issued block by block in turns instead of one after the other
        # (networks.interleaved_forward).  Throughput-neutral on this host (the GPU is saturated either way), so the
        # longer-tested sequential issue order stays the default.
        self.interleave_encoders = bool(tuning.host.interleave)
        unused = [p for m in self.models.values() for n, p in m.named_parameters() if n.startswith("encoder.fc.")]
        self.grad_sync = dp.GradientSynchronizer(self.flat, world_size, never_used=unused,
                                                 segments=[len(list(m.parameters())) for m in self.models.values()])
        if self.opt.train_load_weights_folder is not None:
            self.load_model()

        # API-parity members (trainer.py:180-194); the fused kernel does not need baked pixel grids
        if not self.opt.no_ssim:
            self.ssim = SSIM()
        self.backproject_depth, self.project_3d = {}, {}
        for scale in self.opt.scales:
            h, w = self.opt.height // (2 ** scale), self.opt.width // (2 ** scale)
            self.backproject_depth[scale] = BackprojectDepth(self.batch_size, h, w)
            self.project_3d[scale] = Project3D(self.batch_size, h, w)
        self.photo_options = FD.PhotoOptions(self.opt.min_depth, self.opt.max_depth, self.opt.no_ssim,
                                             self.opt.avg_reprojection, self.opt.gdc_loss_threshold, self.opt.si_var,
                                             si_mode=self._lidar_term()[1])
        self.depth_metric_names = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]
        self.epoch, self.step, self.batch_idx = 0, 0, 0
        self.best = 10.0
        self.start_time = time.time()
        self.set_train()
        self.flat.zero_grad()
        if verbose and rank == 0:
            n = sum(p.numel() for p in self.parameters_to_train)
            print("fusiondepth_amd.Trainer: %d parameters (%.1f MB fp32), accumulating %d steps, single batch size = %d, "
                  "lr %.3g, %d rank(s)" % (n, n * 4 / 1e6, self.accumulate_step, self.batch_size, self.lr, world_size))

    # ---- what the sibling drivers of the reference (completor.py) configure differently -----------------
    def _derived_hparams(self, vram_gib):
        return derived_hparams(self.opt, vram_gib)

    def _encoder_layers(self):
        """(ResNet depth of the depth / beam encoders, of the pose / beam-pose encoders)."""
        return self.opt.num_layers, self.opt.num_layers

    def _lidar_term(self):
        """(scales at which the LiDAR loss is evaluated, fd_photo_cfg.si_mode, name of the entry in ``losses``)
        trainer.py:569-589."""
        if self.opt.trainer_siloss != "true":
            return [], 0, "si_loss"
        return (list(self.opt.scales) if self.opt.trainer_siloss_all_scale else [0]), 0, "si_loss"

    # ------------------------------------------------------------------------------------------------
    def set_train(self):
        """trainer.py:207-211"""
        for m in self.models.values():
            m.train()

    def set_eval(self):
        """trainer.py:213-217"""
        for m in self.models.values():
            m.eval()

    # ------------------------------------------------------------------------------------------------
    def stack_micro_batches(self, micro_batches):
        """Concatenate the accumulated micro-batches along the batch axis (see ``train_step``)."""
        if len(micro_batches) == 1:
            return micro_batches[0]
        out = {}
        for k, v0 in micro_batches[0].items():
            if torch.is_tensor(v0):
                out[k] = torch.cat([mb[k] for mb in micro_batches], 0)
            elif k == "_noise":
                out[k] = [torch.cat([mb[k][s] for mb in micro_batches], 0) for s in range(len(v0))]
            else:
                out[k] = v0
        return out

    def train_step(self, micro_batches):
        """One optimiser step = ``accumulate_step`` micro-batches (trainer.py:237-248: loss/accumulate, backward, step
        every accumulate-th batch).

        With ``stack_microbatches`` (default) the micro-batches run as ONE stacked forward/backward: BatchNorm normalises
        each micro-batch separately (``FD.bn_groups``, running statistics updated in micro-batch order), the SI-log loss is
        evaluated per micro-batch, and the other loss terms are means, so loss = sum_g loss_g / accumulate_step and its
        gradient equal the reference's accumulated values while every kernel sees twice the work per launch.
        Returns the loss dict (device tensors; nothing is synchronised here)."""
        prestacked = isinstance(micro_batches, dict)       # a loader that already delivers the step's images as one batch
        assert prestacked or len(micro_batches) == self.accumulate_step
        if self.stack_microbatches:
            self.grad_sync.arm()
            stacked = micro_batches if prestacked else self.stack_micro_batches(micro_batches)
            outputs, losses = self.process_batch(stacked, groups=self.accumulate_step)
            losses["loss"].backward()
            self._join_side_streams()
            self.batch_idx += self.accumulate_step
        else:
            assert not prestacked, "pre-stacked input needs stack_microbatches"
            losses = None
            for i, inputs in enumerate(micro_batches):
                last = i == self.accumulate_step - 1
                outputs, losses = self.process_batch(inputs)
                loss = losses["loss"] / self.accumulate_step
                if last:
                    self.grad_sync.arm()
                loss.backward()
                self._join_side_streams()
                self.batch_idx += 1
        scale = self.grad_sync.finish() if self.world_size > 1 else 1.0
        self.optimizer_step(scale)
        self._ensure_weight_plan()
        self.step += self.accumulate_step            # the reference counts batches (trainer.py:264), not optimiser steps
        self._last_io = (stacked if self.stack_microbatches else inputs, outputs)
        return losses

    def _ensure_weight_plan(self):
        """After the first full forward + backward every conv weight has its cached kernel-side layouts: collect their
        re-layout work into one device job table, so that from now on Adam is followed by ONE re-layout launch instead of
        ~220 small ones spread over the next step.  (Not capturable: call outside graph capture.)"""
        if FD.weight_plan_needs_rebuild():
            had_plan = FD._WT_PLAN[0] is not None
            if FD.build_weight_plan() > 0 and not had_plan:
                FD.refresh_weight_layouts()

    def optimizer_step(self, grad_scale=1.0):
        """torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8).step(); zero_grad()  as one fused kernel.  The step
        counter and lr are read from device memory so the launch can live inside a captured hipGraph."""
        self.adam_step_count += 1
        FD.adam_step_dev(self.flat.flat_param, self.flat.flat_grad, self.exp_avg, self.exp_avg_sq, self.adam_state,
                         grad_scale=grad_scale)
        self.flat.zero_grad()

    def lr_scheduler_step(self):
        """StepLR(step_size, 0.1).step()  (trainer.py:266): the learning rate drops by 10x every ``scheduler_step_size`` calls."""
        self.scheduler_epochs = getattr(self, "scheduler_epochs", 0) + 1
        if self._graph is None:                 # no captured graph can hold a raw pointer into a retired layout buffer: free them
            FD.release_retired_layouts()
        if self.scheduler_step_size > 0 and self.scheduler_epochs % self.scheduler_step_size == 0:
            self.lr *= 0.1
            self.adam_state[1] = self.lr

    def end_epoch(self):
        """For callers that drive ``train_step`` themselves: advance the epoch counter and the StepLR schedule."""
        self.epoch += 1
        self.lr_scheduler_step()

    # ---- driver loop (trainer.py:219-266, 632-642) -----------------------------------------------------------------------
    def train(self, train_loader=None, val_loader=None):
        """trainer.py:219-228 over ANY iterable of batches in the reference's schema (its DataLoader, a list, a generator
        factory with ``__iter__``): ``num_epochs`` x ``run_epoch``, a checkpoint every ``save_frequency`` epochs."""
        if train_loader is not None:
            self.train_loader = train_loader
        if val_loader is not None:
            self.val_loader = val_loader
        if getattr(self, "train_loader", None) is None:
            raise RuntimeError("Trainer.train: no train_loader (pass one, or set self.train_loader; this package ships no KITTI "
                               "DataLoader - any iterable of reference-schema batches works)")
        self.epoch, self.step = 0, 0
        self.scheduler_epochs = 0              # a second train() on the same object starts its StepLR count afresh
        self.start_time = time.time()
        try:
            self.num_total_steps = len(self.train_loader) * self.opt.num_epochs
        except TypeError:
            self.num_total_steps = 0
        if self.rank == 0:
            self.save_opts()
        for self.epoch in range(self.opt.num_epochs):
            self.run_epoch()
            if (self.epoch + 1) % self.opt.save_frequency == 0 and self.rank == 0:
                self.save_model()

    def _log_due(self, batch_idx, step):
        """trainer.py:252-254: every log_frequency batches during the first 2000 steps, then every 2000 steps."""
        return (batch_idx % self.opt.log_frequency == 0 and step < 2000) or step % 2000 == 0

    def run_epoch(self):
        """trainer.py:230-266.  The reference runs process_batch + backward per batch and steps the optimiser every
        ``accumulate_step`` batches; here the window's batches run as one stacked pass (``train_step``), so logging /
        validation happen once per window, if any batch of the window was due.  A window left unfinished at the end of an epoch is
        dropped, as the reference's ``zero_grad()`` at the start of the next epoch does."""
        self.set_train()
        self.flat.zero_grad()
        window, t0 = [], time.time()
        for batch_idx, inputs in enumerate(self.train_loader):
            if not window:
                t0 = time.time()
            window.append(inputs)
            if len(window) < self.accumulate_step:
                continue
            step0, first = self.step, batch_idx - self.accumulate_step + 1
            losses = self.train_step(window)
            window = []
            if self.rank == 0 and any(self._log_due(first + k, step0 + k) for k in range(self.accumulate_step)):
                loss = float(losses["loss"]) / self.accumulate_step              # the value trainer.py:243 prints (synchronises)
                self.log_time(batch_idx, (time.time() - t0) / self.accumulate_step, loss)
                stacked, outputs = self._last_io
                if "depth_gt" in stacked:
                    # the reference logs the metrics of the batch that was due - the window's last one, not the stacked window
                    n = stacked["depth_gt"].shape[0] // self.accumulate_step if self.stack_microbatches else stacked["depth_gt"].shape[0]
                    last_in = {"depth_gt": stacked["depth_gt"][-n:]}
                    last_out = Outputs.for_options(self.opt, {("disp", 0): outputs[("disp", 0)][-n:]})
                    self.compute_depth_losses(last_in, last_out, losses)
                self.log("train", losses)
                if getattr(self, "val_loader", None) is not None:
                    self.log("val", self.val(self.val_loader))
                    self.set_train()
        self.lr_scheduler_step()

    def log_time(self, batch_idx, duration, loss):
        """trainer.py:632-642."""
        samples_per_sec = self.batch_size / max(duration, 1e-9)
        time_sofar = time.time() - self.start_time
        left = (self.num_total_steps / self.step - 1.0) * time_sofar if self.step > 0 and self.num_total_steps else 0
        if self.verbose:
            print("epoch {:>3} | batch {:>6} | examples/s: {:5.1f} | loss: {:.5f} | time elapsed: {} | time left: {}".format(
                self.epoch, batch_idx, samples_per_sec, loss, _hms(time_sofar), _hms(left)))
        self.last_log_time = dict(epoch=self.epoch, batch=batch_idx, examples_per_s=samples_per_sec, loss=loss)

    def log(self, mode, losses):
        """trainer.py:644-681 without tensorboard / wandb: one JSON line per event in ``<log_path>/<mode>/scalars.jsonl``
        (the image summaries are not written; ``outputs`` keeps everything they were made from)."""
        folder = os.path.join(self.log_path, mode)
        os.makedirs(folder, exist_ok=True)
        rec = {"step": self.step, "epoch": self.epoch, "lr": self.lr}
        for k, v in losses.items():
            rec[k] = float(v)
        with open(os.path.join(folder, "scalars.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")

    # ------------------------------------------------------------------------------------------------
    def train_step_graphed(self, micro_batches):
        """``train_step`` replayed from a captured hipGraph (HIP graphs instead of a tracing compiler): the ~4500
        kernel launches of one optimiser step cost one graph launch on the host.  The first call runs eagerly
        (allocator warm-up), the second captures, later calls copy the new batch into the captured input buffers
        and replay.  With several ranks the forward/backward micro-steps are replayed and the gradient all-reduce +
        Adam run after the graph."""
        self._pose_on_main = True              # the captured step keeps the pose decoder on the capture stream (see predict_poses)
        try:
            return self._train_step_graphed(micro_batches)
        finally:
            self._pose_on_main = False         # direct process_batch / train_step calls afterwards fork their side streams again

    def _train_step_graphed(self, micro_batches):
        if self._graph is None:
            if self.stack_microbatches:
                self._static_in = self.stack_micro_batches(micro_batches)
                if len(micro_batches) == 1:
                    self._static_in = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self._static_in.items()}
            else:
                self._static_in = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in mb.items()} for mb in micro_batches]
            self._last_mbs = micro_batches
            self._graph = "warm"
            # warm up on the stream the capture will use, so that autograd's AccumulateGrad nodes are bound to it
            self._side = torch.cuda.Stream()
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                losses = self._graph_body(self._static_in)
                if self.world_size > 1:
                    self._sync_and_step()
            torch.cuda.current_stream().wait_stream(self._side)
            self.step += self.accumulate_step
            self.batch_idx += self.accumulate_step
            return losses
        if micro_batches is not self._last_mbs:
            self._copy_into_static(micro_batches)
            self._last_mbs = micro_batches
        if self._graph == "warm":
            with torch.cuda.stream(self._side):
                self._ensure_weight_plan()     # layouts valid now; inside the graph Adam is followed by the batched refresh
            if FD._WT_PLAN[0] is None:
                FD.bump_weights_epoch()        # no plan: the captured step must re-derive every weight layout at first use
            FD.sync_late_layouts()             # no event from outside the capture may be waited on inside it
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=self._side):
                self._static_losses = self._graph_body(self._static_in)
            self._graph = g
        # A refresh of the cached weight layouts issued eagerly since the last replay - the optimiser step that follows the graph when
        # world_size > 1, load_model() - puts the large layouts on a side stream behind an event (functional.refresh_weight_layouts).
        # The captured kernels were recorded with "layout ready" and never look at that event: the replay stream waits for it here.
        FD.sync_late_layouts()
        self._graph.replay()
        if self.world_size > 1:
            self._sync_and_step()
        self.step += self.accumulate_step
        self.batch_idx += self.accumulate_step
        return self._static_losses

    def _copy_into_static(self, micro_batches):
        if not self.stack_microbatches:
            for dst, src in zip(self._static_in, micro_batches):
                for k, v in src.items():
                    if torch.is_tensor(v):
                        dst[k].copy_(v)
            return
        for i, mb in enumerate(micro_batches):
            for k, v in mb.items():
                if torch.is_tensor(v):
                    n = v.shape[0]
                    self._static_in[k][i * n:(i + 1) * n].copy_(v)
                elif k == "_noise":
                    for s_, t in enumerate(v):
                        self._static_in[k][s_][i * t.shape[0]:(i + 1) * t.shape[0]].copy_(t)

    def _graph_body(self, static_in):
        if self.stack_microbatches:
            outputs, losses = self.process_batch(static_in, groups=self.accumulate_step)
            losses["loss"].backward()
            self._join_side_streams()
        else:
            losses = None
            for inputs in static_in:
                outputs, losses = self.process_batch(inputs)
                (losses["loss"] / self.accumulate_step).backward()
                self._join_side_streams()
        if self.world_size == 1:
            self.adam_step_count += 1
            FD.adam_step_dev(self.flat.flat_param, self.flat.flat_grad, self.exp_avg, self.exp_avg_sq, self.adam_state)
            self.flat.flat_grad.zero_()
        return {k: v.detach() for k, v in losses.items()}

    def _sync_and_step(self):
        import torch.distributed as dist
        dist.all_reduce(self.flat.flat_grad, op=dist.ReduceOp.SUM)
        self.optimizer_step(1.0 / self.world_size)

    # ------------------------------------------------------------------------------------------------
    def _join_side_streams(self):
        """Backward kernels of the encoder modules run on their forward streams and accumulate parameter gradients in
        place (no AccumulateGrad node => autograd does not sync those streams for us): order everything before the
        optimiser / all-reduce on the current stream."""
        cur = torch.cuda.current_stream()
        for st in self._streams:
            cur.wait_stream(st)
        FD.join_wgrad_streams()

    def _fork(self, idx):
        """Side stream #idx, ordered after everything already queued on the current stream."""
        idx = idx % max(int(tuning.host.n_streams), 1)
        while len(self._streams) <= idx:
            self._streams.append(torch.cuda.Stream())
        st = self._streams[idx]
        st.wait_stream(torch.cuda.current_stream())
        return st

    def _join(self, st, tensors):
        cur = torch.cuda.current_stream()
        cur.wait_stream(st)
        for t in tensors:
            if t is not None:
                t.record_stream(cur)

    def process_batch(self, inputs, val=False, groups=1):
        """trainer.py:268-319 (default separate-pose-encoder path).  ``groups`` > 1: ``inputs`` holds that many micro-batches
        stacked along the batch axis (see ``train_step``).

        The six ResNet passes of a training batch (RGB encoder, beam encoder, and pose / beam-pose encoders for
        frames -1 and +1) are mutually independent; at micro-batch 6 a single pass cannot fill 256 CUs (layer4 has
        720 output pixels), so the four encoder modules are issued on separate HIP streams — inside the captured hipGraph they become
        parallel branches — and joined before the decoders.  Autograd replays each backward on its forward stream,
        so the backward passes overlap the same way."""
        for key, ipt in inputs.items():
            if key != "date" and key != "path" and torch.is_tensor(ipt) and ipt.device != self.device:
                inputs[key] = ipt.to(self.device)
        par = (self.parallel_streams and not val and self.use_pose_net and self.opt.pose_model_type == "separate_resnet"
               and self.num_pose_frames == 2 and len(self._pose_fids()) > 0)
        if self.opt.cat_4beam_to_color:
            enc_in = torch.cat((inputs["color_aug", 0, 0], inputs["4beam"]), 1)
        elif self.opt.cat2start:
            enc_in = torch.cat((inputs["color_aug", 0, 0], inputs["2channel"]), 1)
        else:
            enc_in = inputs["color_aug", 0, 0]
        self._groups = groups
        FD.begin_forward_pass()
        with FD.defer_bn_counters():
            return self._process_batch(inputs, val, groups, par, enc_in)

    def _pose_fids(self):
        """The source frames whose pose comes from the pose network: all but the stereo partner "s" (trainer.py:337, 382)."""
        return [f for f in self.opt.frame_ids[1:] if f != "s"]

    def _process_batch_shared(self, inputs, val):
        """trainer.py:275-283: one depth-encoder pass over ALL frames (batch len(frame_ids) * B, one set of BatchNorm
        statistics), split per frame; the depth decoder sees frame 0, the pose decoder the per-frame features."""
        fids = self.opt.frame_ids
        all_features = self.models["encoder"](torch.cat([inputs[("color_aug", i, 0)] for i in fids]))
        B = inputs[("color_aug", 0, 0)].shape[0]
        features = {k: [f[i * B:(i + 1) * B] for f in all_features] for i, k in enumerate(fids)}
        outputs = Outputs.for_options(self.opt, self.models["depth"](features[0]))
        if self.use_pose_net and not val:
            outputs.update(self.predict_poses(inputs, features))
        losses = {}
        if val:
            self.generate_images_pred(inputs, outputs, [0])
        else:
            self.generate_images_pred(inputs, outputs, self.opt.frame_ids)
            losses = self.compute_losses(inputs, outputs)
        return outputs, losses

    def _process_batch(self, inputs, val, groups, par, enc_in):
        pose_out = None
        if groups > 1 and not par:
            raise NotImplementedError("stacked micro-batches need the separate_resnet pose path; set stack_microbatches=False")
        if self.opt.pose_model_type == "shared":
            return self._process_batch_shared(inputs, val)
        interleave = par and self.interleave_encoders and self.opt.beam_encoder and not self.opt.cat2end
        if par and not interleave:
            pose_out = self._launch_pose_encoders(inputs)          # side streams, joined in predict_poses
        beam_features = None
        if interleave:
            features, beam_features, pose_out = self._encoders_interleaved(inputs, enc_in, groups)
        else:
            if self.opt.beam_encoder and not self.opt.cat2end:
                if par:
                    st = self._fork(0)
                    with torch.cuda.stream(st), FD.bn_groups(groups):
                        beam_features = self._run_module("beam_encoder", inputs["2channel"])
                else:
                    beam_features = self._run_module("beam_encoder", inputs["2channel"])
            with FD.bn_groups(groups):
                features = self._run_module("encoder", enc_in)
            if par and beam_features is not None:
                self._join(self._streams[0], beam_features)
        if self.opt.cat2end:
            outputs = self.models["depth"](features, two_channel=inputs["2channel"])
        elif beam_features is not None:
            if not self.opt.predictive_mask:               # features[0] of both encoders feed the depth decoder's last concatenation only
                FD.mark_single_consumer(features[0], beam_features[0])
            outputs = self.models["depth"](features, beam_features=beam_features)
        else:
            outputs = self.models["depth"](features)
        outputs = Outputs.for_options(self.opt, outputs)
        if self.opt.predictive_mask:                                                          # trainer.py:305-306
            outputs["predictive_mask"] = dict(self.models["predictive_mask"](features))
        if self.use_pose_net and not val:
            outputs.update(self.predict_poses(inputs, features, pose_out))
        losses = {}
        if val:
            self.generate_images_pred(inputs, outputs, [0])
        else:
            if par and tuning.host.smooth_stream and not self._pose_on_main and not torch.cuda.is_current_stream_capturing():
                # the smoothness terms (3 small launches per scale, forward and backward) beside the photometric kernel instead of
                # behind it on the main stream; the beam encoder's stream is idle between its forward and its backward
                st = self._fork(0)
                with torch.cuda.stream(st):
                    outputs["_smooth"] = ([FD.normalized_smooth_loss(outputs[("disp", sc)], inputs[("color", 0, sc)])
                                           for sc in self.opt.scales], st)
            self.generate_images_pred(inputs, outputs, self.opt.frame_ids)
            losses = self.compute_losses(inputs, outputs)
        return outputs, losses

    def _launch_pose_encoders(self, inputs):
        """pose_encoder / beam_encoder_pose for the source frames (trainer.py:336-351), one stream per module.

        The reference runs each module once per source frame (batch B).  Here the frame pairs are stacked along the batch
        axis and run as ONE pass of batch 2B with grouped BatchNorm (``FD.bn_groups(2)``): per-sample arithmetic, the
        per-pass batch statistics and the order of the running-statistics updates are those of the two separate passes,
        but every kernel sees twice the pixels (layer4: 1 440 instead of 720) and half the launches are issued."""
        fids = self._pose_fids()
        G = self._groups
        stack = lambda key: self._stack_pose_inputs(inputs, key)
        res = {}
        st_rgb = self._fork(1)
        with torch.cuda.stream(st_rgb):
            with FD.bn_groups(G * len(fids)):
                pf = self._run_module("pose_encoder", stack("color_aug"))
        bf, st_beam = None, None
        if self.opt.beam_encoder:
            st_beam = self._fork(2)
            with torch.cuda.stream(st_beam):
                with FD.bn_groups(G * len(fids)):
                    bf = self._run_module("beam_encoder_pose", stack("2channel"))
        res["stacked"] = (pf, st_rgb, bf, st_beam)
        self._early_loss_inputs(inputs, st_beam if st_beam is not None else st_rgb)
        return res

    def _early_loss_inputs(self, inputs, st):
        """The identity reprojection losses (trainer.py:515-528) and the tie-break noise (trainer.py:551-552) are functions of the batch
        alone, but the reference - and rounds 1-5 here - evaluate them between the depth decoder and the loss kernel: the one place of
        the step where every stream waits for the main one.  Issued here instead, behind the shortest encoder stream's forward pass;
        ``generate_images_pred`` joins that stream (it is joined for the poses anyway).  Same kernels, same order of random draws."""
        self._loss_pre = None
        o = self.opt
        if (not tuning.host.early_loss_inputs or o.disable_automasking or o.v1_multiscale or torch.cuda.is_current_stream_capturing()
                or not self._multiscale_loss_ok(self._loss_fids())):
            return
        with torch.cuda.stream(st):
            ident0 = self.identity_losses(inputs, 0)
            noise = None
            if inputs.get("_noise") is None:
                B, _, H, W = inputs[("color", 0, 0)].shape
                noise = torch.randn((len(o.scales), B, ident0.shape[1], H, W), device=ident0.device)
        self._loss_pre = (inputs, ident0, noise, st)

    def _loss_fids(self):
        return [f for f in self.opt.frame_ids[1:]]

    _REPLAYED = ("encoder", "beam_encoder", "pose_encoder", "beam_encoder_pose")

    def _run_module(self, name, *tensors):
        """``self.models[name](*tensors)``.  The four ResNet encoders in training mode go through ``replay.TrainReplayable``: after two
        eager steps their forward and backward passes are recorded call sequences behind FIVE autograd nodes each (stem, layer1 .. 4),
        issued by one ``fd_replay`` call per node and direction instead of ~45 + ~90 Python-issued launches (the step's host issue
        time); the Refiner routes its frozen networks through recorded sequences here too (its own override)."""
        net = self.models[name]
        if name in self._REPLAYED and tuning.host.replay_train and net.training and torch.is_grad_enabled() and len(tensors) == 1:
            reps = self.__dict__.setdefault("_train_replays", {})
            rp = reps.get(name)
            if rp is None:
                from .replay import ReplayedEncoder
                rp = reps[name] = ReplayedEncoder(net, "Trainer." + name)
            return rp(tensors[0])
        return net(*tensors)

    def _stack_pose_inputs(self, inputs, key):
        """The (source, target) frame pairs of all source frames stacked along the batch axis, in the reference's pass order:
        for each micro-batch g, for each source frame f  (trainer.py:237-248 + 336-351)."""
        fids = self._pose_fids()
        orders = [(f, 0) if f < 0 else (0, f) for f in fids]
        G = self._groups
        Bg = inputs["color_aug", 0, 0].shape[0] // G
        # one launch assembles AND normalises the stacked tensor (rounds 2-4: 8 slice copies + the encoder's normalisation pass at the
        # head of the longest streams of the step)
        first = inputs[key, orders[0][0], 0]
        C = first.shape[1]
        pieces = []
        for g in range(G):
            for k, o in enumerate(orders):
                for j, i in enumerate(o):
                    pieces.append((inputs[key, i, 0][g * Bg:(g + 1) * Bg], (g * len(orders) + k) * Bg, j * C))
        if first.is_cuda and (first.shape[2] * first.shape[3]) % 4 == 0 and all(p[0].is_contiguous() and p[0].data_ptr() % 16 == 0 for p in pieces):
            out = FD.stack_normalize(pieces, G * len(orders) * Bg, 2 * C)
            out._fd_normalized = True
            return out
        out = torch.empty((G * len(orders) * Bg, 2 * C) + tuple(first.shape[2:]), device=first.device, dtype=first.dtype)
        for t, img0, ch0 in pieces:
            out[img0:img0 + Bg, ch0:ch0 + C].copy_(t)
        return out

    def _encoders_interleaved(self, inputs, enc_in, groups):
        """All four encoder modules of the step, each on its own stream, issued block by block in turns
        (networks.interleaved_forward) -> (features, beam_features, pose_out for predict_poses)."""
        nf = len(self._pose_fids())
        st_rgb, st_beam, st_lidar = self._fork(1), self._fork(2), self._fork(0)
        with torch.cuda.stream(st_rgb):
            pose_in = self._stack_pose_inputs(inputs, "color_aug")
        with torch.cuda.stream(st_beam):
            beam_pose_in = self._stack_pose_inputs(inputs, "2channel")
        jobs = [(self.models["pose_encoder"], pose_in, st_rgb, groups * nf),
                (self.models["beam_encoder_pose"], beam_pose_in, st_beam, groups * nf),
                (self.models["beam_encoder"], inputs["2channel"], st_lidar, groups),
                (self.models["encoder"], enc_in, None, groups)]
        pf, bf, beam_features, features = networks.interleaved_forward(jobs)
        self._join(st_lidar, beam_features)
        self._early_loss_inputs(inputs, st_beam)
        return features, beam_features, {"stacked": (pf, st_rgb, bf, st_beam)}

    def predict_poses(self, inputs, features, precomputed=None):
        """trainer.py:321-388.  ``precomputed``: encoder features already launched on side streams.  The stereo partner "s"
        has no predicted pose (trainer.py:337, 382); with ``--pose_model_type shared`` ``features`` maps frame id -> the depth
        encoder's feature list of that frame (trainer.py:330-331, 376-377)."""
        outputs = {}
        shared = self.opt.pose_model_type == "shared"
        fids = self._pose_fids()
        if self.num_pose_frames == 2:
            stacked = None
            pose_stream = None
            if precomputed is not None:
                pf, st, bf, st2 = precomputed["stacked"]
                # The pose decoder (a dozen small launches forward, ~60 backward incl. autograd's slicing glue) stays on the pose
                # encoder's stream: autograd replays a node on its forward stream, so the decoder's backward - created last, hence
                # replayed first - no longer sits on the main stream in front of the depth decoder's backward (tuning.host.pose_stream).
                if (tuning.host.pose_stream and self.parallel_streams and not self._pose_on_main
                        and not torch.cuda.is_current_stream_capturing()):
                    pose_stream = st
                    if bf is not None:
                        st.wait_stream(st2)
                        for t in bf:
                            if t is not None:              # features[0] of a pose encoder is not materialised (stem_feature_needed)
                                t.record_stream(st)
                else:
                    self._join(st, pf)
                    if bf is not None:
                        self._join(st2, bf)
            with torch.cuda.stream(pose_stream) if pose_stream is not None else contextlib.nullcontext():
                heads = None
                if precomputed is not None:
                    fused = tuning.host.fused_pose_head and len(fids) <= 4
                    res = self.models["pose"]([pf], beam_inputs=[bf] if bf is not None else None, raw=fused)   # batch = len(fids) * B
                    if fused:           # slices, concatenations and pose matrices of every frame pair: one launch each way
                        nf, G = len(fids), self._groups
                        heads = FD.pose_head(res, G, nf, res.shape[0] // (nf * G), [f_i < 0 for f_i in fids])
                    else:
                        stacked = res
                for k, f_i in enumerate(fids):
                    if heads is not None:
                        (outputs[("cam_T_cam", 0, f_i)], outputs[("axisangle", 0, f_i)], outputs[("translation", 0, f_i)]) = heads[k]
                        continue
                    order = (f_i, 0) if f_i < 0 else (0, f_i)                     # temporal order (trainer.py:338-346)
                    if stacked is not None:
                        nf, G = len(fids), self._groups
                        Bq = stacked[0].shape[0] // (nf * G)                 # rows are ordered (micro-batch g, frame k, sample)
                        if G == 1:
                            axisangle, translation = stacked[0][k * Bq:(k + 1) * Bq], stacked[1][k * Bq:(k + 1) * Bq]
                        else:
                            sl = [slice((g * nf + k) * Bq, (g * nf + k + 1) * Bq) for g in range(G)]
                            axisangle = torch.cat([stacked[0][q] for q in sl], 0)
                            translation = torch.cat([stacked[1][q] for q in sl], 0)
                    elif shared:
                        axisangle, translation = self.models["pose"]([features[i] for i in order])
                    else:
                        pose_inputs = torch.cat([inputs["color_aug", i, 0] for i in order], 1)
                        if self.opt.pose_model_type == "separate_resnet":
                            pose_inputs = [self.models["pose_encoder"](pose_inputs)]
                        if self.opt.beam_encoder and self.opt.pose_model_type == "separate_resnet":
                            beam = torch.cat([inputs["2channel", i, 0] for i in order], 1)
                            beam_inputs = [self.models["beam_encoder_pose"](beam)]
                            axisangle, translation = self.models["pose"](pose_inputs, beam_inputs=beam_inputs)
                        else:
                            axisangle, translation = self.models["pose"](pose_inputs)
                    outputs[("axisangle", 0, f_i)] = axisangle
                    outputs[("translation", 0, f_i)] = translation
                    outputs[("cam_T_cam", 0, f_i)] = transformation_from_parameters(axisangle[:, 0], translation[:, 0],
                                                                                    invert=(f_i < 0))
            if pose_stream is not None:
                self._join(pose_stream, list(outputs.values()))
        else:
            if shared:
                pose_inputs = [features[i] for i in self.opt.frame_ids if i != "s"]
            else:
                pose_inputs = torch.cat([inputs[("color_aug", i, 0)] for i in self.opt.frame_ids if i != "s"], 1)
                if self.opt.pose_model_type == "separate_resnet":
                    pose_inputs = [self.models["pose_encoder"](pose_inputs)]
            axisangle, translation = self.models["pose"](pose_inputs)
            for i, f_i in enumerate(self.opt.frame_ids[1:]):
                if f_i != "s":
                    outputs[("axisangle", 0, f_i)] = axisangle
                    outputs[("translation", 0, f_i)] = translation
                    outputs[("cam_T_cam", 0, f_i)] = transformation_from_parameters(axisangle[:, i], translation[:, i])
        return outputs

    # ------------------------------------------------------------------------------------------------
    def identity_losses(self, inputs, source_scale=0):
        """trainer.py:515-528: the identity reprojection losses do not depend on the pyramid scale when
        source_scale == 0, so they are computed once per batch ([B,NF,H,W]) instead of once per scale."""
        target = inputs[("color", 0, source_scale)]
        fids = self.opt.frame_ids[1:]
        B, _, H, W = target.shape
        ident = torch.empty(B, len(fids), H, W, device=target.device)
        for i, f in enumerate(fids):
            FD.reprojection_loss_map(inputs[("color", f, source_scale)], target, not self.opt.no_ssim, out=ident[:, i:i + 1])
        if self.opt.avg_reprojection:
            ident = ident.mean(1, keepdim=True)
        return ident

    def generate_images_pred(self, inputs, outputs, frame_ids):
        """trainer.py:425-474 fused with the per-pixel part of compute_losses (trainer.py:509-567, 577-589).

        Stores, per scale s, outputs[("photo", s)] = (to_optimise.mean(), si_loss) and
        "identity_selection/s"; and, if ``materialize_outputs``, the reference's ("depth",0,s), ("sample",f,s),
        ("color",f,s), ("color_identity",f,s)."""
        fids = [f for f in frame_ids[1:]]
        if not fids:                                  # validation: depth only (trainer.py:311-312)
            for scale in self.opt.scales:
                disp = outputs[("disp", scale)]
                if not self.opt.v1_multiscale:
                    disp = FD.bilinear_upsample(disp, (self.opt.height, self.opt.width))
                outputs[("depth", 0, scale)] = disp_to_depth(disp, self.opt.min_depth, self.opt.max_depth)[1]
            return
        automask = not self.opt.disable_automasking
        si_scales = self._lidar_term()[0]
        pre, self._loss_pre = getattr(self, "_loss_pre", None), None
        early_noise = None
        if pre is not None and pre[0] is inputs and automask and not self.opt.v1_multiscale:     # issued beside the encoders (_early_loss_inputs)
            ident0, early_noise = pre[1], pre[2]
            self._join(pre[3], [ident0, early_noise])
        else:
            ident0 = self.identity_losses(inputs, 0) if (automask and not self.opt.v1_multiscale) else None
        noise_in = inputs.get("_noise")               # injected tie-break noise (tests); else drawn like trainer.py:551-552
        if self._multiscale_loss_ok(fids):
            # default configuration: ALL scales in one launch that also produces the gradients (csrc/photometric_ms.hip)
            scales = list(self.opt.scales)
            B, _, H, W = inputs[("color", 0, 0)].shape
            noise = None
            if ident0 is not None:
                if noise_in is not None:
                    noise = [noise_in[s] for s in scales]
                else:
                    noise = list(early_noise if early_noise is not None else
                                 torch.randn((len(scales), B, ident0.shape[1], H, W), device=ident0.device))
            lidar = [i for i, s in enumerate(scales) if s in si_scales]
            photo, si, sel = FD.photo_loss_ms(
                [outputs[("disp", s)] for s in scales], [outputs[("cam_T_cam", 0, f)] for f in fids], inputs[("K", 0)],
                inputs[("inv_K", 0)], [inputs[("color", f, 0)] for f in fids], inputs[("color", 0, 0)], ident0, noise,
                inputs["4beam"] if lidar else None, lidar, self.photo_options, getattr(self, "_groups", 1))
            for i, scale in enumerate(scales):
                outputs[("photo", scale)] = (photo[i], si[i])
                if automask:
                    outputs[("sel", scale)] = (sel[i], ident0.shape[1])
            return
        pose_of = lambda f: inputs["stereo_T"] if f == "s" else outputs[("cam_T_cam", 0, f)]     # trainer.py:444-447
        for scale in self.opt.scales:
            src_s = scale if self.opt.v1_multiscale else 0
            target = inputs[("color", 0, src_s)]
            ident = None
            if automask:
                ident = ident0 if ident0 is not None else self.identity_losses(inputs, src_s)
            noise = None
            if ident is not None:
                noise = noise_in[scale] if noise_in is not None else torch.randn(ident.shape, device=ident.device)
            beam = inputs["4beam"] if (scale in si_scales and src_s == 0) else None
            if self.opt.pose_model_type == "posecnn":
                # trainer.py:450-460: PoseCNN translations are rescaled by the mean inverse depth of this scale
                disp_up = outputs[("disp", scale)]
                if not self.opt.v1_multiscale:
                    disp_up = FD.bilinear_upsample(disp_up, (self.opt.height, self.opt.width))
                inv_depth = disp_to_depth(disp_up, self.opt.min_depth, self.opt.max_depth)[0]       # 1 / depth
                mean_inv_depth = FD.spatial_mean(inv_depth, 1.0)                                       # [B,1]
                Ts = [inputs["stereo_T"] if f == "s" else
                      transformation_from_parameters(outputs[("axisangle", 0, f)][:, 0],
                                                     outputs[("translation", 0, f)][:, 0] * mean_inv_depth[:, None, :],
                                                     f < 0) for f in fids]
            else:
                Ts = [pose_of(f) for f in fids]
            Ts = [T if T.shape[0] == target.shape[0] else T.expand(target.shape[0], 4, 4).contiguous() for T in Ts]
            srcs = [inputs[("color", f, src_s)] for f in fids]
            mask = weighting = None
            if self.opt.predictive_mask and not automask:
                # trainer.py:530-541: the reprojection losses are multiplied by the predicted per-frame mask, and a
                # 0.2 * BCE(mask, 1) term keeps the mask from collapsing to 0
                mask = outputs["predictive_mask"][("disp", scale)]
                if not self.opt.v1_multiscale:
                    mask = FD.bilinear_upsample(mask, (self.opt.height, self.opt.width))
                weighting = 0.2 * (-torch.clamp(torch.log(mask), min=-100.0)).mean()      # nn.BCELoss()(mask, ones)
            photo, si, sel, depth, sample, color = FD.photo_loss(
                outputs[("disp", scale)], Ts, inputs[("K", src_s)], inputs[("inv_K", src_s)], srcs, target, ident, noise,
                beam, self.photo_options, self.materialize_outputs, getattr(self, "_groups", 1), mask=mask)
            if weighting is not None:
                photo = photo + weighting
            if scale in si_scales and src_s != 0:
                # --v1_multiscale: the photometric terms live at the scale's own resolution, the LiDAR term at full resolution
                # (trainer.py:577-589 upsamples disp before it).  A second pass of the kernel at full resolution supplies it;
                # only its LiDAR output is used, so its photometric inputs are placeholders.
                f0 = fids[0]
                si = FD.photo_loss(outputs[("disp", scale)], [Ts[0]], inputs[("K", 0)], inputs[("inv_K", 0)],
                                   [inputs[("color", f0, 0)]], inputs[("color", 0, 0)], None, None, inputs["4beam"],
                                   self.photo_options, False, getattr(self, "_groups", 1))[1]
                beam = inputs["4beam"]
            outputs[("photo", scale)] = (photo, si if beam is not None else None)
            if automask:
                outputs[("sel", scale)] = (sel, ident.shape[1])
            if self.materialize_outputs:
                outputs[("depth", 0, scale)] = depth
                for i, f in enumerate(fids):
                    outputs[("sample", f, scale)] = sample[i]
                    outputs[("color", f, scale)] = color[i]
                    if automask:
                        outputs[("color_identity", f, scale)] = inputs[("color", f, src_s)]

    def _multiscale_loss_ok(self, fids):
        """The fused all-scales kernel covers the default loss configuration (two source frames, SSIM, per-frame minimum, one
        pose per frame shared by the scales, no materialised warps); every flag variant keeps the per-scale kernels."""
        if not tuning.host.photo_ms:
            return False
        o = self.opt
        return (FD.photo_ms_supported(self.photo_options, len(fids), self.materialize_outputs) and not o.v1_multiscale
                and o.pose_model_type != "posecnn" and not o.predictive_mask and "s" not in fids and 1 <= len(o.scales) <= 4
                and all(o.height % (2 ** s) == 0 and o.width % (2 ** s) == 0 for s in o.scales))

    def compute_reprojection_loss(self, pred, target):
        """trainer.py:476-488 (stand-alone, differentiable w.r.t. ``pred`` through the SSIM kernel)."""
        l1_loss = torch.abs(target - pred).mean(1, True)
        if self.opt.no_ssim:
            return l1_loss
        return 0.85 * FD.ssim(pred, target).mean(1, True) + 0.15 * l1_loss

    def compute_losses(self, inputs, outputs):
        """trainer.py:490-596: per scale  loss = min-reprojection mean + smoothness/2^s ; total += loss + si_loss."""
        losses = {}
        scales = list(self.opt.scales)
        lidar_name = self._lidar_term()[2]
        photo, si, smooth = [], [], []
        pre = outputs.pop("_smooth", None)
        if pre is not None:
            self._join(pre[1], pre[0])
        for i, scale in enumerate(scales):
            p, s_ = outputs[("photo", scale)]
            photo.append(p); si.append(s_)
            smooth.append(pre[0][i] if pre is not None else
                          FD.normalized_smooth_loss(outputs[("disp", scale)], inputs[("color", 0, scale)]))
        if scales == list(range(len(scales))) and len(scales) == self.num_scales and len(scales) <= 4:
            per_scale, total = FD.combine_losses(photo, smooth, si, self.opt.disparity_smoothness)     # one kernel each way
            for scale in scales:
                losses["loss/{}".format(scale)] = per_scale[scale]
                if si[scale] is not None:
                    losses["loss/{}{}".format(lidar_name, scale)] = si[scale]
            losses["loss"] = total
            return losses
        total_loss = 0
        for i, scale in enumerate(scales):
            loss = photo[i] + self.opt.disparity_smoothness * smooth[i] / (2 ** scale)
            total_loss = total_loss + loss
            losses["loss/{}".format(scale)] = loss
            if si[i] is not None:
                total_loss = total_loss + si[i]
                losses["loss/{}{}".format(lidar_name, scale)] = si[i]
        losses["loss"] = total_loss / self.num_scales
        return losses

    # ------------------------------------------------------------------------------------------------
    def compute_depth_losses(self, inputs, outputs, losses, accumulate=False):
        """trainer.py:598-630 — monitoring metrics vs depth_gt (Garg crop, median scaling)."""
        depth_pred = outputs[("depth", 0, 0)].detach()
        gt_h, gt_w = inputs["depth_gt"].shape[2:]
        depth_pred = torch.clamp(FD.bilinear_upsample(depth_pred, (gt_h, gt_w)), 1e-3, 80) if gt_h >= depth_pred.shape[2] \
            else torch.clamp(torch.nn.functional.interpolate(depth_pred, [gt_h, gt_w], mode="bilinear", align_corners=False), 1e-3, 80)
        depth_gt = inputs["depth_gt"]
        mask = depth_gt > 0
        crop = torch.zeros_like(mask)
        crop[:, :, 153:371, 44:1197] = 1
        mask = mask * crop
        gt, pred = depth_gt[mask], depth_pred[mask]
        pred = torch.clamp(pred * (torch.median(gt) / torch.median(pred)), min=1e-3, max=80)
        errs = FD.depth_errors(gt, pred)
        for i, metric in enumerate(self.depth_metric_names):
            v = errs[i].detach().cpu().numpy()
            losses[metric] = losses.get(metric, 0.0) + v if accumulate else v

    def val_metrics(self, batches):
        """trainer.py:390-409: mean monitoring metrics over an iterable of batches (depth only, eval-mode BN)."""
        self.set_eval()
        losses = {m: 0.0 for m in self.depth_metric_names}
        n = 0
        with torch.no_grad():
            for inputs in batches:
                outputs, _ = self.process_batch(inputs, val=True)
                if "depth_gt" in inputs:
                    self.compute_depth_losses(inputs, outputs, losses, accumulate=True)
                n += 1
        for m in self.depth_metric_names:
            losses[m] /= max(n, 1)
        self.set_train()
        return losses

    def val(self, batches, save_best=True):
        """trainer.py:390-423: validation metrics + best-checkpoint bookkeeping - a new best de/abs_rel is remembered and
        saved as ``weights_best`` and, below 0.080, also as ``weights_absrel<round(1000 * abs_rel)>``.  The folders written
        are left in ``self.last_saved``."""
        losses = self.val_metrics(batches)
        self.last_saved = []
        if losses["de/abs_rel"] < self.best:
            self.best = float(losses["de/abs_rel"])
            if save_best:
                self.last_saved.append(self.save_model("best"))
                absrel = round(float(losses["de/abs_rel"]) * 1000)
                if absrel < 80:
                    self.last_saved.append(self.save_model("absrel{}".format(absrel)))
        return losses

    # ------------------------------------------------------------------------------------------------
    def save_model(self, name=None):
        """trainer.py:694-715: one state_dict per network under models/weights_<epoch|name>/ (+ height/width/
        use_stereo in encoder.pth) and adam.pth."""
        tag = "weights_{}".format(self.epoch if name is None else name)
        folder = os.path.join(self.log_path, "models", tag)
        os.makedirs(folder, exist_ok=True)
        for model_name, model in self.models.items():
            to_save = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            if model_name == "encoder":
                to_save["height"], to_save["width"], to_save["use_stereo"] = self.opt.height, self.opt.width, self.opt.use_stereo
            torch.save(to_save, os.path.join(folder, "{}.pth".format(model_name)))
        torch.save(self.optimizer_state_dict(), os.path.join(folder, "adam.pth"))
        return folder

    def optimizer_state_dict(self):
        """``torch.optim.Adam.state_dict()`` layout (what trainer.py:714-715 writes), so that ``adam.pth`` interchanges with the
        reference: per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` in ``parameters_to_train`` order (the reference builds
        that list in the same network order, trainer.py:66-129), one parameter group carrying the current learning rate."""
        state = {}
        step = torch.tensor(float(self.adam_step_count))
        if self.adam_step_count > 0:
            avg, sq = self.exp_avg.detach().cpu(), self.exp_avg_sq.detach().cpu()
            for i, p in enumerate(self.flat.params):
                o, n = self.flat.offsets[i], p.numel()
                state[i] = {"step": step.clone(), "exp_avg": avg[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": sq[o:o + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "initial_lr": self.learning_rate,
                 "params": list(range(len(self.flat.params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, st):
        """Inverse of ``optimizer_state_dict``; also reads the flat layout round 1 of this package wrote.  Restores the moments,
        the step count (bias correction) and the learning rate (StepLR decays already taken) on the host AND in the device-side
        ``adam_state`` the Adam kernel reads."""
        if "state" in st and "param_groups" in st:
            n_params = len(self.flat.params)
            groups = st["param_groups"]
            listed = sum(len(g["params"]) for g in groups)
            if listed != n_params:
                raise RuntimeError("adam.pth holds %d parameters, this trainer has %d (different network set?)" % (listed, n_params))
            steps = []
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            for i, entry in st["state"].items():
                i = int(i)
                p, o = self.flat.params[i], self.flat.offsets[i]
                if tuple(entry["exp_avg"].shape) != tuple(p.shape):
                    raise RuntimeError("adam.pth: moment %d has shape %s, parameter has %s" % (i, tuple(entry["exp_avg"].shape), tuple(p.shape)))
                self.exp_avg[o:o + p.numel()].copy_(entry["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + p.numel()].copy_(entry["exp_avg_sq"].reshape(-1))
                steps.append(int(float(entry["step"])))
            if steps and min(steps) != max(steps):
                raise RuntimeError("adam.pth: per-parameter step counts differ (%d..%d); the flat Adam kernel keeps one" % (min(steps), max(steps)))
            self.adam_step_count = steps[0] if steps else 0
            self.lr = float(groups[0]["lr"])
        elif "exp_avg" in st:
            if st["exp_avg"].numel() != self.exp_avg.numel():
                raise RuntimeError("adam.pth: %d moments for %d parameters" % (st["exp_avg"].numel(), self.exp_avg.numel()))
            self.exp_avg.copy_(st["exp_avg"]); self.exp_avg_sq.copy_(st["exp_avg_sq"])
            self.adam_step_count = int(st["step"])
            self.lr = float(st.get("lr", self.lr))
        else:
            raise RuntimeError("adam.pth: unknown layout (keys %s)" % sorted(st))
        self.adam_state[0] = float(self.adam_step_count)
        self.adam_state[1] = self.lr

    def load_model(self):
        """trainer.py:717-746: weights of ``models_to_load`` (+ the two LiDAR encoders when ``--beam_encoder``, :726-728) copied in
        place into the flat parameter buffer, then the Adam state if ``adam.pth`` is there."""
        folder = os.path.expanduser(self.opt.train_load_weights_folder)
        assert os.path.isdir(folder), "Cannot find folder {}".format(folder)
        names = list(self.opt.models_to_load)
        if self.opt.beam_encoder:
            names += [n for n in ("beam_encoder", "beam_encoder_pose") if n not in names]
        for n in names:
            if n not in self.models:
                raise KeyError("models_to_load: this trainer has no network %r (has %s)" % (n, sorted(self.models)))
            path = os.path.join(folder, "{}.pth".format(n))
            if not os.path.isfile(path):
                raise FileNotFoundError("models_to_load: %s is missing" % path)
            model_dict = self.models[n].state_dict()
            pretrained = torch.load(path, map_location="cpu")
            with torch.no_grad():
                for k, v in pretrained.items():
                    if k in model_dict:
                        model_dict[k].copy_(v)              # in place: parameters stay views of the flat buffer
        FD.bump_weights_epoch()
        FD.invalidate_frozen_layouts()
        FD.refresh_weight_layouts()       # a captured step holds no per-conv re-layout launches: refresh the cached copies now
        adam = os.path.join(folder, "adam.pth")
        if os.path.isfile(adam):
            self.load_optimizer_state_dict(torch.load(adam, map_location="cpu"))
        elif self.verbose:
            print("Cannot find Adam weights so Adam is randomly initialized")

    def save_opts(self):
        """trainer.py:683-692."""
        folder = os.path.join(self.log_path, "models")
        os.makedirs(folder, exist_ok=True)
        with open(os.path.join(folder, "opt.json"), "w") as f:
            json.dump(self.opt.__dict__.copy(), f, indent=2)
