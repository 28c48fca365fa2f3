"""HIP-backed ``Refiner`` — the training step of the reference's ``refiner.py`` (BASELINE.json config 5)
on the same kernels as the trainer's hot path.

The trained depth / pose networks are frozen (eval-mode BatchNorm, ``torch.no_grad`` for the depth branch); a second decoder,
``refine2d_decoder`` = DepthDecoder(road=True, catxy, deep), is trained.  Per scale it receives the coarse disparity rescaled
to the sparse LiDAR's metric scale (median ratio inside the crop), the pseudo-3D ``Cat_xy`` coordinates and the max-pooled
2-channel LiDAR map (refiner.py:316-348); its output goes through the trainer's fused warp + SSIM/L1 + min-reprojection
kernel, whose masked scale-invariant log term is pointed at the dense GDC depth ``inputs["inf_gdc"]`` with the refiner's
constants (refiner.py:557-563: mask lower bound 1e-3, no 26x / 100x scaling, factor 10 * gdc_loss_weight [* 4]).

Reference map: refiner.py:299-382 process_batch, :383-448 predict_poses, :487-541 generate_images_pred, :557-563 siloss,
:592-693 compute_losses, :80-167 model set / Adam.  Glue that is not arithmetic of the hot path (2x2 ceil-mode max-pooling of
three small maps, the masked medians, the bilinear down-sampling of 1/depth) uses ATen ops on the GPU.
"""
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import dp
from . import functional as FD
from . import tuning
from . import networks
from .layers import disp_to_depth
from .trainer import Outputs, Trainer, derived_hparams

REFINER_MODEL_ORDER = ["encoder", "beam_encoder", "beam_encoder_pose", "depth", "pose_encoder", "pose", "refine2d_decoder"]


def _lookahead(loader):
    """(batch, next batch or None) pairs of an iterable."""
    it = iter(loader)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


class Refiner(Trainer):
    """Same method names as the reference's ``Refiner``; batches are dicts with the reference's keys (+ ``"inf_gdc"``)."""

    def __init__(self, options, device=None, rank=0, world_size=1, verbose=True):
        self.opt = options
        self.verbose = verbose
        if self.opt.no_cuda or not torch.cuda.is_available():
            raise RuntimeError("fusiondepth_amd.Refiner needs an MI355X: there is no CPU path (use oracle/ for CPU checks)")
        self.opt.clone_gdc, self.opt.refine_2d = True, True                                   # refiner.py:29-30
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.index is not None:
            torch.cuda.set_device(self.device)           # every raw-stream launch below targets the current device
        self.rank, self.world_size = rank, world_size
        self.materialize_outputs = False
        self.log_path = os.path.join(self.opt.log_dir, self.opt.model_name)
        vram = torch.cuda.get_device_properties(self.device).total_memory / 1024 ** 3
        hp = derived_hparams(self.opt, vram)                                                  # refiner.py:32-45 (same rule)
        self.opt.num_epochs = hp["num_epochs"]
        self.accumulate_step, self.learning_rate = hp["accumulate_step"], hp["learning_rate"]
        self.scheduler_step_size, self.batch_size = hp["scheduler_step_size"], hp["micro_batch"]
        self.eval_scales = self.opt.scales
        assert self.opt.height % 32 == 0 and self.opt.width % 32 == 0, "'height' / 'width' must be multiples of 32"
        assert self.opt.frame_ids[0] == 0, "frame_ids must start with 0"
        if self.opt.train_entire_net:
            # refiner.py:306-313 computes `features` only under `if not self.opt.train_entire_net`, so the reference itself
            # stops with an UnboundLocalError at the first batch: there is no behaviour to reproduce
            raise NotImplementedError("--train_entire_net: the reference's Refiner.process_batch (refiner.py:306-313) never "
                                      "computes the encoder features on that path (UnboundLocalError at the first batch)")
        if self.opt.use_stereo or self.opt.predictive_mask or self.opt.pose_model_type != "separate_resnet" or \
                self.opt.v1_multiscale or not self.opt.beam_encoder:
            raise NotImplementedError("Refiner: only the default path (frozen nets, separate_resnet pose net, beam encoder, "
                                      "full-resolution sampling) is implemented")
        self.num_scales = len(self.opt.scales)
        self.num_input_frames = len(self.opt.frame_ids)
        self.num_pose_frames = 2 if self.opt.pose_model_input == "pairs" else self.num_input_frames
        self.use_pose_net = True

        m = {}                                                                                # refiner.py:80-160
        m["encoder"] = networks.ResnetEncoder(self.opt.num_layers, False, cat4beam_to_color=self.opt.cat_4beam_to_color,
                                              cat2channel=self.opt.cat2start)
        m["beam_encoder"] = networks.ResnetEncoder(self.opt.num_layers, False, beam_encoder=True)
        m["beam_encoder_pose"] = networks.ResnetEncoder(self.opt.num_layers, False, num_input_images=self.num_pose_frames,
                                                        beam_encoder=True)
        m["depth"] = networks.DepthDecoder(m["encoder"].num_ch_enc, self.opt.scales, cat2end=self.opt.cat2end)
        m["pose_encoder"] = networks.ResnetEncoder(self.opt.num_layers, False, num_input_images=self.num_pose_frames)
        m["pose"] = networks.PoseDecoder(m["pose_encoder"].num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
        m["refine2d_decoder"] = networks.DepthDecoder(m["encoder"].num_ch_enc, self.opt.scales, road=True,
                                                      catxy=(self.opt.catxy == "true"), deep=(self.opt.refine2d_deep == "true"))
        self.models = {k: m[k].to(self.device) for k in REFINER_MODEL_ORDER}
        self._load_pretrained()                                                                # refiner.py:56-60, 84-152
        if world_size > 1:
            dp.broadcast_module_state(self.models.values())
        self.parameters_to_train = list(self.models["refine2d_decoder"].parameters())          # refiner.py:148-160
        for k, net in self.models.items():
            if k != "refine2d_decoder":
                for p in net.parameters():
                    p.requires_grad_(False)          # frozen: no data gradient is propagated into them either

        self.flat = dp.FlatParameters(self.parameters_to_train)
        FD.evict_dead_weight_layouts()
        FD.enable_weight_cache(self.parameters_to_train)
        FD.enable_direct_grad(self.parameters_to_train)
        # the refine decoder is the step's serial chain (one batch-6 launch at a time): its weight gradients - leaves of the backward
        # graph - run on a side stream beside the following layers' data gradients, like the depth decoder's in the Trainer
        if "refine2d_decoder" in tuning.host.side_wgrad:
            FD.enable_side_wgrad(self.parameters_to_train)
        # the frozen stage-1 networks: kernel-side weight layouts derived once, not on every call
        FD.enable_weight_cache([p for k, net in self.models.items() if k != "refine2d_decoder" for p in net.parameters()], frozen=True)
        self.exp_avg = torch.zeros_like(self.flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat_param)
        self.adam_step_count = 0
        self.lr = self.learning_rate
        self.adam_state = torch.tensor([0.0, self.lr], device=self.device)
        self._graph, self._streams = None, []
        self._replays = {}
        self._frozen_stream, self._prefetched = None, None
        self.parallel_streams = tuning.host.refiner_streams
        self.stack_microbatches = False
        self._groups = 1
        self.grad_sync = dp.GradientSynchronizer(self.flat, world_size)
        self.photo_options = FD.PhotoOptions(self.opt.min_depth, self.opt.max_depth, self.opt.no_ssim, self.opt.avg_reprojection,
                                             self.opt.gdc_loss_threshold, self.opt.si_var, si_depth_scale=1.0,
                                             si_beam_scale=1.0, si_lo=1e-3)
        self.depth_metric_names = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]
        self.epoch, self.step, self.batch_idx = 0, 0, 0
        self.best = 10.0
        self.start_time = time.time()
        self.set_train()
        self.flat.zero_grad()
        if verbose:
            n = sum(p.numel() for p in self.parameters_to_train)
            print("fusiondepth_amd.Refiner: training refine2d_decoder, %d parameters (%.1f MB fp32), lr %.3g" % (n, n * 4 / 1e6, self.lr))

    def _load_pretrained(self):
        """refiner.py:56-60 + the ``load_state_dict`` after every constructor (:84-152): the frozen depth / pose networks come
        from ``--refine_load_weights_folder`` (stage-1 ``Trainer.save_model`` output; the encoder file also carries height /
        width / use_stereo, filtered by key like the reference does), ``refine2d_decoder.pth`` is optional (resume).
        Without a folder (synthetic benchmarks / unit tests only) the networks keep their random initialisation."""
        folder = self.opt.refine_load_weights_folder
        if folder is None:
            if self.verbose:
                print("fusiondepth_amd.Refiner: no --refine_load_weights_folder; frozen networks keep their random initialisation")
            return
        folder = os.path.expanduser(folder)
        assert os.path.isdir(folder), "Cannot find a folder at {}".format(folder)
        for name, net in self.models.items():
            path = os.path.join(folder, "{}.pth".format(name))
            if not os.path.isfile(path):
                if name == "refine2d_decoder":
                    continue
                raise FileNotFoundError("refine_load_weights_folder: %s is missing" % path)
            own = net.state_dict()
            loaded = torch.load(path, map_location="cpu")
            missing = [k for k in own if k not in loaded]
            if missing and name != "encoder":
                raise RuntimeError("%s: missing keys %s" % (path, missing[:4]))
            with torch.no_grad():
                for k, v in loaded.items():
                    if k in own:
                        own[k].copy_(v)
        FD.bump_weights_epoch()
        FD.invalidate_frozen_layouts()

    def set_train(self):
        """refiner.py:80-160: the depth / pose networks stay in eval mode; only the refine decoder trains."""
        for k, net in self.models.items():
            net.train() if k == "refine2d_decoder" else net.eval()

    # ------------------------------------------------------------------------------------------------
    def refine_inputs(self, inputs, outputs):
        """refiner.py:316-348: median scaling of the coarse depth to the LiDAR returns, scaled disparity, Cat_xy, pooled 2-channel
        map - all scales in four libfdhip launches (``fd_refine_inputs``, csrc/refine.hip: ``torch.median`` of a boolean selection as
        a radix select, no full-resolution intermediates).  Rounds 2-4 built this from ~60 ATen launches per step (nanmedian = a
        sort per median, max_pool2d, interpolate, cat); ``refine_inputs_aten`` keeps that form for the A/B in scripts/."""
        opt = self.opt
        scales = list(opt.scales)
        disps = [outputs[("disp", s)] for s in scales]
        maps = FD.refine_inputs(disps, inputs["4beam"], inputs["2channel"], [inputs[("inv_K", s)] for s in scales], opt.height, opt.width,
                                opt.min_depth, opt.max_depth, catxy=(opt.catxy == "true"), pool_disp0=(opt.refine_a0 == "true"))
        return {("disp", s): m for s, m in zip(scales, maps)}

    def refine_inputs_aten(self, inputs, outputs):
        """The round 2-4 form of ``refine_inputs`` (ATen calls); kept for timing comparisons only."""
        opt = self.opt
        beam, two_cha = inputs["4beam"], inputs["2channel"]
        disp_0 = outputs[("disp", 0)]
        res = {}
        mask = beam > 0
        crop = torch.zeros_like(mask)
        crop[:, :, 78:190, 23:617] = 1
        mask = mask * crop
        nan = torch.full((), float("nan"), device=beam.device)
        masked_median = lambda x: torch.nanmedian(torch.where(mask, x, nan))
        beam_med = masked_median(beam * 100.0)
        for scale in opt.scales:
            if opt.refine_a0 != "true":
                disp = outputs[("disp", scale)]
            else:
                disp = disp_0
                disp_0 = F.max_pool2d(disp_0, 2, ceil_mode=True)
            disp640 = FD.bilinear_upsample(disp, (opt.height, opt.width)) if disp.shape[2] != opt.height else disp
            depth = disp_to_depth(disp640, opt.min_depth, opt.max_depth)[1]
            depth = depth * (beam_med / masked_median(depth))
            scaled_disp = (F.interpolate(1 / depth, disp.shape[2:], mode="bilinear", align_corners=False) - 0.01) / 9.9
            if scale != 0:
                two_cha = F.max_pool2d(two_cha, 2, ceil_mode=True)
            if opt.catxy == "true":
                for _ in range(scale):
                    depth = F.max_pool2d(depth, 2, ceil_mode=True)
                xyz = FD.cat_xy(depth, inputs[("inv_K", scale)])
                res[("disp", scale)] = torch.cat([scaled_disp, xyz, two_cha], 1)
            else:
                res[("disp", scale)] = torch.cat([scaled_disp, two_cha], 1)
        return res

    def _to_device(self, inputs):
        for key, ipt in inputs.items():
            if torch.is_tensor(ipt) and ipt.device != self.device:
                inputs[key] = ipt.to(self.device)

    def _frozen_block(self, inputs, val):
        """Everything of refiner.py:299-330 that involves only the batch and frozen networks (no parameter that is trained, no
        autograd graph): the frozen forward passes, the refine decoder's input maps and the poses -> (features, beam_features, outputs)."""
        # The four frozen encoders are independent and, at the Refiner's batch of 6, none of them fills 256 CUs: like the
        # Trainer they run on one HIP stream per module, the two pose passes (frames -1 / +1) stacked into one pass per module.
        # Eval-mode BatchNorm is per sample, so stacking changes nothing; no autograd graph is recorded for any of them.
        par = self.parallel_streams and self.use_pose_net and not val and self.num_pose_frames == 2
        want_poses = self.use_pose_net and not val
        with torch.no_grad():
            # (Replaying this frozen section from a hipGraph was tried: 348 vs 356 images/s - hipGraphLaunch on this ROCm costs the host
            # as much per kernel node as the eager launches it replaces)
            features, beam_features, depth, poses = self._frozen_forward(inputs, par, want_poses)
            outputs = Outputs.for_options(self.opt, depth)
            outputs.update(self.refine_inputs(inputs, outputs))
            outputs.update(poses)
        return features, beam_features, outputs

    def prefetch_frozen(self, next_inputs):
        """Issue the frozen block of the NEXT batch now, on its own stream, so that the GPU runs it beside this batch's refine-decoder
        forward / loss / backward (a serial chain of batch-6 launches that leaves most CUs idle).  Nothing in the block reads a trained
        parameter, so its results are those of issuing it at the start of the next ``train_step`` - which then picks them up (matched
        by the identity of the ``next_inputs`` dict) instead of recomputing.  The stream waits for everything queued on the current
        stream up to here, so tensors of ``next_inputs`` that the caller is still producing on it are complete."""
        self._to_device(next_inputs)
        cur = torch.cuda.current_stream()
        if self._frozen_stream is None:
            self._frozen_stream = torch.cuda.Stream()
        st = self._frozen_stream
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            frozen = self._frozen_block(next_inputs, False)
            done = torch.cuda.Event()
            done.record(st)
        self._prefetched = (next_inputs, frozen, done)

    def _take_prefetched(self, inputs):
        pre, self._prefetched = self._prefetched, None
        if pre is None or pre[0] is not inputs:
            return None
        _, frozen, done = pre
        cur = torch.cuda.current_stream()
        cur.wait_event(done)
        features, beam_features, outputs = frozen
        for t in list(features) + list(beam_features or []) + list(outputs.values()):
            if torch.is_tensor(t):
                t.record_stream(cur)          # allocated on the prefetch stream's pool, read (and released) on this one
        return frozen

    def process_batch(self, inputs, val=False, frozen=None):
        """refiner.py:299-382 (train_entire_net=False).  ``frozen``: the result of ``_frozen_block`` for this batch, if the caller
        already has it (``train_step`` with a prefetched block)."""
        self._to_device(inputs)
        FD.begin_forward_pass()
        if frozen is None:
            frozen = self._frozen_block(inputs, val)
        features, beam_features, outputs = frozen
        losses = {"loss": 0.0}
        n_iter = self.opt.refine_iter
        for it in range(n_iter):
            offset = self.models["refine2d_decoder"](features, beam_features=beam_features, depth_maps=outputs,
                                                     tanh=self.opt.refine_offset)
            for s in self.opt.scales:
                outputs[("disp", s)] = offset[("disp", s)]
            self.generate_images_pred(inputs, outputs, [0] if val else self.opt.frame_ids)
            if not val:
                gama = (1.0 if n_iter == 1 else self.opt.refine_iter_gama) ** (n_iter - it)
                losses = self.compute_losses(inputs, outputs, losses, gama=gama)
        return outputs, losses

    def _early_loss_inputs(self, inputs, st):
        """(Trainer's hoist of the identity losses / noise: the Refiner's own ``generate_images_pred`` evaluates them per refine iteration)"""
        self._loss_pre = None

    def _frozen_forward(self, inputs, par, want_poses):
        """The part of refiner.py:299-330 that involves only frozen networks: depth / beam encoders, depth decoder and (training)
        the pose networks -> (features, beam_features, {("disp", s)}, {cam_T_cam / axisangle / translation}).  Call under no_grad."""
        pose_out = None
        if par:
            pose_out = self._launch_pose_encoders(inputs)
            st = self._fork(0)
            with torch.cuda.stream(st):
                beam_features = self._run_module("beam_encoder", inputs["2channel"])
        features = self._run_module("encoder", inputs["color_aug", 0, 0])
        if par:
            self._join(st, beam_features)
        else:
            beam_features = self._run_module("beam_encoder", inputs["2channel"])
        if self.opt.refine_depthnet_with_beam == "true":
            depth = dict(self._run_module("depth", *features, *beam_features))
        else:
            depth = dict(self._run_module("depth", *features))
        poses = self.predict_poses(inputs, features, pose_out) if want_poses else {}
        return features, beam_features, depth, poses

    def _run_module(self, name, *tensors):
        """A frozen stage-1 network under no_grad: its libfdhip calls are recorded once per input signature and replayed by ONE
        ``fd_replay`` call afterwards (replay.py; ~45 launches per ResNet-18 encoder, ~40 per depth decoder: the step was host-bound).
        ``depth`` takes the encoder features (+ the LiDAR encoder's) as a flat argument list."""
        net = self.models[name]
        if name == "depth":
            n = len(net.num_ch_enc)
            call = lambda *f: net(list(f[:n]), beam_features=list(f[n:])) if len(f) > n else net(list(f))
        else:
            call = lambda x: list(net(x))
        if name == "refine2d_decoder" or torch.is_grad_enabled() or net.training:
            return call(*tensors)
        rp = self._replays.get(name)
        if rp is None:
            from .replay import Replayable
            rp = self._replays[name] = Replayable(call, lambda: list(net.parameters()) + list(net.buffers()), name="Refiner." + name)
        return rp(*tensors)

    def generate_images_pred(self, inputs, outputs, frame_ids):
        """refiner.py:487-541 fused with the per-pixel part of compute_losses; the SI term is the GDC loss (compute_losses)."""
        fids = [f for f in frame_ids[1:]]
        if not fids:
            for scale in self.opt.scales:
                disp = FD.bilinear_upsample(outputs[("disp", scale)], (self.opt.height, self.opt.width))
                outputs[("depth", 0, scale)] = disp_to_depth(disp, self.opt.min_depth, self.opt.max_depth)[1]
            return
        automask = not self.opt.disable_automasking
        ident = self.identity_losses(inputs, 0) if automask else None
        noise_in = inputs.get("_noise")
        if self._multiscale_loss_ok(fids):             # all scales in one launch (csrc/photometric_ms.hip)
            scales = list(self.opt.scales)
            noise = None
            if ident is not None:
                noise = [noise_in[s] for s in scales] if noise_in is not None else \
                    list(torch.randn((len(scales),) + tuple(ident.shape), device=ident.device))
            gdc = [i for i, s in enumerate(scales) if (not self.opt.gdc_loss_only_on_scale_0) or s == 0]
            photo, si, sel = FD.photo_loss_ms(
                [outputs[("disp", s)] for s in scales], [outputs[("cam_T_cam", 0, f)] for f in fids], inputs[("K", 0)],
                inputs[("inv_K", 0)], [inputs[("color", f, 0)] for f in fids], inputs[("color", 0, 0)], ident, noise,
                inputs["inf_gdc"] if gdc else None, gdc, self.photo_options, 1)
            for i, scale in enumerate(scales):
                outputs[("photo", scale)] = (photo[i], si[i])
                if automask:
                    outputs[("sel", scale)] = (sel[i], ident.shape[1])
            return
        for scale in self.opt.scales:
            noise = None
            if ident is not None:
                noise = noise_in[scale] if noise_in is not None else torch.randn(ident.shape, device=ident.device)
            use_gdc = (not self.opt.gdc_loss_only_on_scale_0) or scale == 0
            Ts = [outputs[("cam_T_cam", 0, f)] for f in fids]
            srcs = [inputs[("color", f, 0)] for f in fids]
            photo, si, sel, depth, sample, color = FD.photo_loss(
                outputs[("disp", scale)], Ts, inputs[("K", 0)], inputs[("inv_K", 0)], srcs, inputs[("color", 0, 0)], ident, noise,
                inputs["inf_gdc"] if use_gdc else None, self.photo_options, self.materialize_outputs, 1)
            outputs[("photo", scale)] = (photo, si if use_gdc else None)
            if automask:
                outputs[("sel", scale)] = (sel, ident.shape[1])

    def compute_losses(self, inputs, outputs, losses, gama=1.0, frame_ids=None):
        """refiner.py:592-693.  The fused kernel's SI term is 0.1 * sqrt(var); the refiner's is 10 * sqrt(var) * weight [* 4]."""
        total = 0
        for scale in self.opt.scales:
            photo, si = outputs[("photo", scale)]
            smooth = FD.normalized_smooth_loss(outputs[("disp", scale)], inputs[("color", 0, scale)])
            loss = photo + self.opt.disparity_smoothness * smooth / (2 ** scale)
            total = total + loss
            losses["loss/gama{}_scale{}".format(gama, scale)] = loss
            if si is not None:
                gdc_loss = si * (100.0 * self.opt.gdc_loss_weight * (4.0 if self.opt.gdc_loss_only_on_scale_0 else 1.0))
                total = total + gdc_loss
                losses["loss/gdc_scale{}".format(scale)] = gdc_loss
        total = total / self.num_scales
        losses["loss"] = losses["loss"] + total * gama
        return losses

    def run_epoch(self):
        """refiner.py:264-297: one optimiser step per batch, the trainer's logging / validation cadence, StepLR at the end."""
        self.set_train()
        for batch_idx, (inputs, next_inputs) in enumerate(_lookahead(self.train_loader)):
            t0 = time.time()
            step0 = self.step
            losses = self.train_step(inputs, next_inputs)
            if self.rank == 0 and self._log_due(batch_idx, step0):
                self.log_time(batch_idx, time.time() - t0, float(losses["loss"]))
                if "depth_gt" in inputs:
                    self.compute_depth_losses(inputs, self._last_outputs, losses)
                self.log("train", {k: v for k, v in losses.items() if torch.is_tensor(v) or np.ndim(v) == 0})
                if getattr(self, "val_loader", None) is not None:
                    self.log("val", self.val(self.val_loader))
                    self.set_train()
        self.lr_scheduler_step()

    def train_step(self, inputs, next_inputs=None):
        """One optimiser step of refiner.py:272-278: zero_grad, backward, step for every batch (the reference's Refiner never
        accumulates, whatever --batch_size is).  With several ranks the refine decoder's gradient is all-reduced (mean).
        ``next_inputs``: the batch of the following call, if the caller has it (``run_epoch`` reads one batch ahead): its frozen
        forward passes are issued first and overlap this step (``prefetch_frozen``, tuning.host.refiner_prefetch)."""
        self.grad_sync.arm()
        self._to_device(inputs)
        frozen = self._take_prefetched(inputs)
        if frozen is None:                                     # first step of a loop, or a caller that does not look ahead
            frozen = self._frozen_block(inputs, False)
        if next_inputs is not None and tuning.host.refiner_prefetch:
            self.prefetch_frozen(next_inputs)                  # queued in front of this step's refine-decoder work
        outputs, losses = self.process_batch(inputs, frozen=frozen)
        losses["loss"].backward()
        FD.join_wgrad_streams()
        scale = self.grad_sync.finish() if self.world_size > 1 else 1.0
        self.optimizer_step(scale)
        self._ensure_weight_plan()
        self.step += 1
        self.batch_idx += 1
        self._last_outputs = outputs
        return losses
