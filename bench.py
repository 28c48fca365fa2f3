#!/usr/bin/env python
"""Benchmark of the FusionDepth training hot path on MI355X (contract: see the task statement / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (one rank per GPU, RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), and as a plain ``python bench.py --gpus N``, which re-launches itself
through torch.distributed.run on 127.0.0.1 with N ranks.  It refuses to run when fewer than N devices are visible and checks
``ranks_seen == N`` and the RCCL backend before printing.

One "step" = one optimiser step of the reference trainer at --batch_size 12 (ResNet-18, 640x192, 4-beam):
trainer.py:28-41 turns that into 2 accumulated micro-batches of 6 images per process, then Adam.  Inputs: a pool of distinct
synthetic scene batches (fusiondepth_amd.synthetic.make_scene_batch: a ground-truth depth field, frames -1 / +1 rendered
through it, LiDAR returns sampled from it), a fresh one per step, all resident in HBM before the timed region starts;
everything inside the timed region is the real work: 6 ResNet passes, decoder, pose decoder, fused photometric/LiDAR loss at 4
scales, full backward, gradient all-reduce (N>1), Adam.  The run fails (exit code 3) if the final loss is not finite.
Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# analytic forward MACs per image of the conv stack, R18 / R50 at 640x192
CONV_GFLOP_FWD_BWD = {(18, 192, 640): 187.3, (50, 192, 640): 406.1, (18, 320, 1024): 499.4, (50, 320, 1024): 1083.0}
LOSS_BYTES_PER_PIXEL = 343.7          # compulsory fwd+bwd HBM traffic of the fused loss path per image
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
# what the arithmetic is, next to `dtype` (an undisclosed precision change would void the line).  Every tensor is fp32;
# products and sums are fp32 (v_mfma_f32_32x32x2_f32 / 16x16x4: exact fp32, Winograd transforms in fp32) EXCEPT, for layers with >= 64
# channels on both sides: (i) the convolutions no Winograd form exists for (csrc/conv_limb.hip), in all three directions - the 1x1
# stride-1 layers (ResNet-50 bottlenecks; on the ResNet-18 configurations only the PoseDecoder's 512 -> 256 squeeze) and the stride-2
# layers (3x3 layerN.0 of every ResNet; the 1x1 downsample branch where its launch fills the chip); (ii) the Winograd WEIGHT gradient
# of the 3x3 stride-1 layers (k_wgrad_wino_limb: transforms in fp32, then the matrix loop).  There each fp32 operand is split exactly
# into three bf16 limbs and the product is six bf16 MFMAs accumulated in fp32 - error against float64 equal to the f32 kernels' on every
# shape (tests/test_gpu_limb.py, test_winograd_weight_gradient_split_precision_vs_float64), all float64-anchored parity bounds
# unchanged.  Every Winograd forward / data-gradient kernel (the roofline probe among them) is fp32 throughout: the split-precision slab
# kernel (fd_tuning.wino_fwd_limb) is off by default because one full-size backward bound does not hold with it.
ARITH_NOTE = ("fp32 storage, fp32 accumulate; products fp32-exact (f32 MFMA) except, with >= 64 channels: convolutions without a Winograd form "
              "(1x1 stride-1; 3x3 stride-2 and large 1x1 stride-2; forward, data and weight gradient) and the Winograd weight gradient of "
              "the 3x3 layers: bf16x3 split, six products, fp32 accumulate (fp32-equal error vs float64; fd_tuning.limb_1x1 = "
              "limb_conv = wino_wgrad_limb = 0 restores the f32 kernels)")
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--num_layers", type=int, default=18)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--batch_size", type=int, default=12, help="per-process --batch_size of the reference trainer")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python on the four module streams (the default since round 5)")
    ap.add_argument("--graph", action="store_true", help="replay a captured hipGraph of the step instead (slower on ROCm 7.2; rounds 1-4 chose between the two "
                    "during the untimed warm-up; the replay never won)")
    ap.add_argument("--no_stack", action="store_true", help="run the accumulated micro-batches sequentially (reference order) "
                    "instead of as one stacked pass with grouped BatchNorm")
    ap.add_argument("--no_roofline", action="store_true")
    ap.add_argument("--share_device", action="store_true", help="testing aid: with --gpus N on a box with fewer devices, let the N "
                    "ranks share them over gloo (FD_DIST_BACKEND=gloo) instead of refusing; such a line is not a scaling measurement")
    ap.add_argument("--pool", type=int, default=0, help="distinct pre-generated step batches (default: one per step incl. warm-up, "
                    "at most 48)")
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps steps each (every window bracketed by barrier + "
                    "synchronize); value / ms_per_step are the MEDIAN window's, min / max are reported beside them")
    ap.add_argument("--no_other_configs", action="store_true", help="skip the short runs of the other BASELINE configurations "
                    "(ResNet-50 batch 8, 1024x320 batch 8, Refiner, Completor) that the 1-GPU line carries as `other_configs`")
    ap.add_argument("--other_steps", type=int, default=10, help="timed steps per window of each `other_configs` entry (3 windows, median "
                                                              "reported; 5 untimed steps before them)")
    ap.add_argument("--_other", default=None, help=argparse.SUPPRESS)      # child mode: run ONE `other_configs` entry, print its JSON
    ap.add_argument("--probe_only", action="store_true", help="run only the roofline probes (no training steps) and print their "
                    "JSON: the command profiled for profiles/*probe_kernel_stats*.md, so that rocprofv3's per-kernel average "
                    "covers the probe launches alone")
    return ap.parse_args()


def graph_time_ms(fn, launches=20, replays=5):
    """Average device time of one fn() call: `launches` calls are captured into a hipGraph and replayed, timed with
    HIP events on the capture/replay stream, so host launch overhead is excluded (kernel time only)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()                                           # warm-up: allocator, hipFuncSetAttribute, autograd nodes
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(launches):
                fn()
        g.replay()
        torch.cuda.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(replays):
            g.replay()
        end.record()
        torch.cuda.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    return start.elapsed_time(end) / (launches * replays)


def roofline_probes(args, tr, batch):
    """Live per-kernel measurements: the dominant MFMA conv shape and the fused loss kernels."""
    from fusiondepth_amd import functional as FD
    out = {}
    B = tr.batch_size
    # dominant kernel: the 3x3 stride-1 convolutions of the ResNet trunks - since round 4 k_conv_wino2p_dma, F(2x2, 3x3) with all 16
    # components in one workgroup (17 % of the step's kernel time, the largest single kernel; rounds 1-3: k_conv_wino, F(2, 3) along x).
    # Probe = layer1's 64->64 conv at H/4 x W/4, forward (the data-gradient launch is the same kernel with transformed flipped
    # weights).  `achieved` divides the ALGORITHMIC flops of the convolution (2*Cout*Cin*9 per output pixel) by the
    # launch time; the kernel itself issues 4/9 of them as MFMA work (16 multiplies per 2x2 outputs instead of 36),
    # `mfma_flop_per_launch` - so `frac` is a Winograd fraction, the matrix-pipe utilisation is `mfma_pipe_frac`.
    # In the training step four such streams run concurrently; the probe runs alone.
    h8, w8 = args.height // 4, args.width // 4
    Bc = B * (tr.accumulate_step if tr.stack_microbatches else 1)      # what the step launches: the stacked micro-batches
    x = torch.randn(Bc, 64, h8, w8, device="cuda")
    w = torch.randn(64, 64, 3, 3, device="cuda") * 0.03
    w._fd_cache_id = -1                                  # as in the step: the (tap, channel)-major weight copy is cached
    with torch.no_grad():
        ms = graph_time_ms(lambda: FD.conv2d(x, w, None, 1, 1))
    flops = 2.0 * Bc * h8 * w8 * 64 * 64 * 9
    # HBM bytes per launch: rocprofv3 --pmc passes cannot run inside this process, so the number comes from the committed passes of
    # this exact kernel and shape (scripts/pmc_probe.sh) - and only while the kernel source is the one that was profiled.
    traffic, traffic_src = None, None
    from fusiondepth_amd import tuning as _tuning
    two_p = _tuning.get_lib()["wino_fwd_2dp_min_wgs"] > 0 and Bc * (h8 // 2) * (w8 // 2) // 64 >= _tuning.get_lib()["wino_fwd_2dp_min_wgs"] and h8 % 2 == 0
    executed = 4.0 / 9.0 if two_p else 2.0 / 3.0
    for name in ("round6_pmc_probe_wino.json", "round5_pmc_probe_wino.json", "round4_pmc_probe_wino.json"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
        src = os.path.join(ROOT, "fusiondepth_amd", "csrc", "conv_wino.hip")
        sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
        if (Bc, h8, w8) != (12, 48, 160):
            break
        if pmc.get("source_sha256") not in (None, sha):
            print("[bench] profiles/%s was measured on a different conv_wino.hip (sha256 %s..., now %s...): roofline.traffic "
                  "left null - re-run scripts/pmc_probe.sh" % (name, pmc["source_sha256"][:12], sha[:12]), file=sys.stderr, flush=True)
            break
        traffic, traffic_src = pmc["traffic_bytes_per_launch"], name
        break
    out["roofline"] = {"bound": "mfma", "kernel": "%s (ResNet layer1 conv 3x3 64->64 @%dx%d, batch %d = the stacked micro-batches, alone on the "
                       "GPU; Winograd %s: algorithmic flops, %s of them executed)"
                       % ("k_conv_wino2p_dma" if two_p else "k_conv_wino", h8, w8, Bc, "F(2x2,3x3)" if two_p else "F(2,3)", "4/9" if two_p else "2/3"),
                       "achieved": flops / (ms * 1e-3) / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                       "frac": flops / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, "traffic": traffic,
                       "traffic_unit": "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, profiles/%s)" % traffic_src,
                       "us_per_launch": ms * 1e3, "flop_per_launch": flops, "mfma_flop_per_launch": flops * executed,
                       "mfma_pipe_frac": flops * executed / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS}
    # fused loss path: the multi-scale kernel (forward + gradients of the four scales in one launch) at the batch the step
    # launches (stacked micro-batches, SI-log statistics per micro-batch).  Reported against the HBM roofline as the north star
    # asks; the kernel is bound by VALU issue (profiles/round2_pmc_loss.md), not by HBM.
    H, W = args.height, args.width
    po = FD.PhotoOptions()
    B, G = Bc, (tr.accumulate_step if tr.stack_microbatches else 1)
    tgt = batch[("color", 0, 0)]
    srcs = [batch[("color", -1, 0)], batch[("color", 1, 0)]]
    ident = tr.identity_losses(batch, 0)
    I = torch.eye(4, device="cuda").repeat(B, 1, 1)
    I[:, 0, 3] = 0.05
    disps = [torch.rand(B, 1, H >> s, W >> s, device="cuda").mul_(0.1).add_(0.02).requires_grad_(True) for s in range(4)]
    noise = list(torch.randn(4, B, 2, H, W, device="cuda"))          # one draw per scale, as trainer.py:551-552
    zero = [torch.zeros((), device="cuda") for _ in range(4)]          # the smoothness terms are separate kernels (not timed here)

    def loss_fwd_bwd():
        # what Trainer.generate_images_pred + compute_losses launch for the four scales: projection matrices, the fused
        # all-scales kernel (value + unit gradients), its scalar reduction, the loss combination, and on the way back the
        # gradient scaling + upsample adjoint launch and the projection-matrix adjoints
        photo, si, _ = FD.photo_loss_ms(disps, [I, I], batch[("K", 0)], batch[("inv_K", 0)], srcs, tgt, ident, noise,
                                        batch["4beam"], (0, 1, 2, 3), po, G)
        FD.combine_losses(photo, zero, si, 1e-3)[1].backward()
    ms = graph_time_ms(loss_fwd_bwd, launches=5)
    byts = LOSS_BYTES_PER_PIXEL * H * W * B
    loss_traffic, loss_traffic_src = None, None
    sha = hashlib.sha256(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fusiondepth_amd", "csrc",
                                           "photometric_ms.hip"), "rb").read()).hexdigest()
    for name in ("round5_pmc_loss.json", "round4_pmc_loss.json", "round3_pmc_loss.json", "round2_pmc_loss.json"):
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)
        if not os.path.exists(pmc_path):
            continue
        pmc = json.load(open(pmc_path))
        if pmc.get("source_sha256") == sha and pmc.get("batch") == B and (H, W) == (192, 640):
            loss_traffic, loss_traffic_src = pmc["traffic_bytes_per_launch"], name
            break
        print("[bench] profiles/%s was measured on a different photometric_ms.hip / batch / size: roofline_loss_path.traffic "
              "left null unless an older record matches - re-run scripts/pmc_loss_ms.sh" % name, file=sys.stderr, flush=True)
    out["roofline_loss_path"] = {"bound": "hbm", "kernel": "k_photo_ms (4 scales, forward + unit gradients) + k_photo_ms_fin + "
                                 "k_photo_ms_bwd (+ projection-matrix and loss-combination launches), batch %d" % B,
                                 "achieved": byts / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                 "frac": byts / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": loss_traffic,
                                 "traffic_unit": "bytes/launch (FETCH_SIZE x2 + WRITE_SIZE of the three kernels, profiles/%s)"
                                                 % loss_traffic_src, "us_per_launch": ms * 1e3, "bytes_per_launch": byts}
    if loss_traffic is not None:
        # The bound that binds: this path is vector-issue-bound, not HBM-bound.  The instruction count is a
        # property of the (unchanged, sha-checked) kernel at this batch and size - profiles/round3_pmc_loss.md: SQ_INSTS_VALU of
        # k_photo_ms<true,true>, 102.4 M wave-instructions = 6.55 G lane-instructions per launch; pure issue time at the measured
        # per-instruction rates 145.5 us (profiles/round5_pmc_loss.md).  Peak = 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz.
        lane_instr, issue_us, peak_slots = 102.4e6 * 64, 145.5, 256 * 4 * 32 * 2.4e9
        out["roofline_loss_path"]["valu"] = {
            "bound": "vector issue", "lane_instructions_per_launch": lane_instr, "peak_lane_slots_per_s": peak_slots,
            "achieved_lane_instructions_per_s": lane_instr / (ms * 1e-3), "frac": lane_instr / (ms * 1e-3) / peak_slots,
            "issue_us_at_measured_rates": issue_us, "issue_efficiency": issue_us / (ms * 1e3),
            "note": "k_photo_ms executes 1 110 lane-instructions per pixel-visit; `frac` = lane-instructions / s over one lane-slot per "
                    "lane and clock (an fp32 add / mul / fma issues a wave in 2.6 - 3.3 cycles, DPP / compare / convert in 4.3 - 5: the "
                    "mix's own ceiling is 0.45 of the slot peak); `issue_efficiency` = the kernel's pure issue time over the whole path's "
                    "time (k_photo_ms alone: 0.55).  0.40 of the HBM roofline would need the path in 158 us - below the 145.5 us of "
                    "issue plus the 47 us of the path's other launches: unreachable for this arithmetic in fp32 (profiles/round5_pmc_loss.md)"}
    return out


def cpu_baseline(args, budget_s=60.0, warm=3, timed_steps=10, threads=None):
    """Reference-equivalent CPU step (oracle = the restatement proven equal to the imported reference), bounded sample:
    BASELINE.json configs[0]: ResNet-18, 640x192, batch 2, fp32, on the host cores this process may use."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import inputs as gin
    from oracle import trainer as OT
    from oracle import scatter as OS
    try:
        host = len(os.sched_getaffinity(0))
    except AttributeError:
        host = os.cpu_count() or 1
    B, H, W = 2, args.height, args.width
    opt = OT.default_opt(height=H, width=W, batch_size=B, num_layers=args.num_layers)
    ot = OT.OracleTrainer(opt, seed=0)
    inp, rng = gin.batch_inputs(7, B, H, W)
    roi = (max(int(round(76 * H / 192)), 2), min(int(round(190 * H / 192)), H - 2), 2, W - 2)
    two = np.stack([np.stack(OS.scatter_2channel_c(inp["4beam"][b, 0].numpy(), roi)) for b in range(B)])
    for f in (0, -1, 1):
        inp[("2channel", f, 0)] = torch.from_numpy(two)
    inp["2channel"] = torch.from_numpy(two)
    def run(n_threads, n_warm, n_timed, budget):
        torch.set_num_threads(n_threads)
        times, t_start = [], time.time()
        for i in range(n_warm + n_timed):
            t0 = time.time()
            ot.micro_step({k: v.clone() for k, v in inp.items()})
            times.append(time.time() - t0)
            if time.time() - t_start > budget and len(times) >= n_warm + 1:
                break
        return times[n_warm:]

    # torch's CPU kernels do not scale to every logical core of a GPU host (measured on the 2 x 64-core / 256-thread box of this
    # pool: 0.82 s per step on 32 threads, 2.0 s on 64, 6.3 s on 128, minutes on 256), so the baseline runs at the thread count
    # that gives the reference path its BEST throughput, found by a short sweep, and reports that count as `cores`.
    if threads is None:
        cand = sorted({c for c in (8, 16, 32, 64) if c <= host} or {host})
        sweep = {}
        for c in cand:
            sweep[c] = float(np.median(run(c, 1, 1, 20.0)))
            if len(sweep) > 1 and sweep[c] > 1.5 * min(sweep.values()):
                break                          # past the knee: more threads only get slower
        threads = min(sweep, key=sweep.get)
    else:
        sweep = {}
    timed = run(threads, warm, timed_steps, budget_s)
    step = float(np.median(timed))
    # ... and on ONE thread (the scalar-port figure of the contract): the trainer, its buffers and torch's kernels are warm from the
    # run above; a single-thread step of this configuration takes ~2.6 s on the GPU boxes' hosts
    one = run(1, 1, 3, 15.0)
    torch.set_num_threads(threads)
    one_s = float(np.median(one))
    one_thread = {"value": B / one_s, "unit": "images/s", "cores": 1, "kind": "port",
                  "sample": "%d timed optimiser steps (1 warm-up) of the same oracle trainer on 1 thread, after the %d-thread run above; median "
                            "%.2f s/step" % (len(one), threads, one_s)}
    return one_thread, {"value": B / step, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "%d timed optimiser steps after %d warm-up steps of the oracle trainer (ResNet-%d, %dx%d, batch %d, fp32, "
                      "torch CPU) on %d threads of a %d-thread host - the fastest of the sweep %s (s/step); median %.2f s/step, "
                      "min %.2f, max %.2f" % (len(timed), warm, args.num_layers, W, H, B, threads, host,
                                              {k: round(v, 2) for k, v in sweep.items()}, step, min(timed), max(timed))}


def _short_run(step, n, warm=5, windows=3):
    """-> (stats, last return value): ``windows`` timed windows of ``n`` steps each after ``warm`` untimed steps, synchronize on both
    sides of every window; stats carry the MEDIAN window's seconds per step ("dt") with the fastest / slowest beside it.  One window
    of five steps (round 4) let a single stall decide the number: the driver's ResNet-50 entry read 53 ms where six builder runs said
    42."""
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    out, win = None, []
    for _ in range(max(windows, 1)):
        t = time.perf_counter()
        for _ in range(n):
            out = step()
        torch.cuda.synchronize()
        win.append((time.perf_counter() - t) / n)
    med = sorted(win)[len(win) // 2]
    return {"dt": med, "windows_ms_per_step": [1e3 * w for w in win], "ms_per_step_min": 1e3 * min(win), "ms_per_step_max": 1e3 * max(win),
            "steps": n, "windows": len(win), "warmup": warm, "reported": "median window"}, out


def _timing_fields(st, images_per_step):
    dt = st["dt"]
    return {"value": images_per_step / dt, "unit": "images/s", "ms_per_step": 1e3 * dt, "steps": st["steps"], "windows": st["windows"],
            "warmup": st["warmup"], "reported": st["reported"], "windows_ms_per_step": st["windows_ms_per_step"],
            "value_min": images_per_step / (1e-3 * st["ms_per_step_max"]), "value_max": images_per_step / (1e-3 * st["ms_per_step_min"])}


class _flop_tally:
    """Counts the direct-equivalent convolution flops the wrapped steps issue (functional.CONV_FLOP_TALLY): the MFMA fraction of
    the configurations that have no analytic table (Refiner: frozen encoders forward only + the refine decoder; Completor)."""

    def __enter__(self):
        from fusiondepth_amd import functional as FD, tuning
        self.FD, self.host = FD, tuning.host
        FD.CONV_FLOP_TALLY = self.t = [0.0]
        # the counter sits in the Python wrappers: the counted (untimed) step issues every network eagerly - a recorded call sequence
        # (replay.py) goes to the library in one C call and would not be seen (round 6's first lines under-reported the Refiner /
        # Completor fractions by the replayed networks' share)
        self.saved = (self.host.replay_train, self.host.replay_frozen)
        self.host.replay_train = self.host.replay_frozen = False
        return self

    def __exit__(self, *exc):
        self.FD.CONV_FLOP_TALLY = None
        self.host.replay_train, self.host.replay_frozen = self.saved

    def flops(self):
        return self.t[0]


OTHER_CONFIGS = ("r50_640x192_b8", "r18_1024x320_b8", "refiner_640x192", "completor_1216x352")


def other_configs(args):
    """The other BASELINE.json configurations on this one GPU, so that the driver's record witnesses them too (VERDICT round 3,
    item 4): config 3 (ResNet-50, 640x192, batch 8), one rank's share of config 4 (ResNet-18, 1024x320, batch 8), config 5 (Refiner
    step at 640x192; Completor step at its 1216x352 resolution).  EACH IN ITS OWN PROCESS (round 5): a training job is a process, and
    inside the headline's process - behind its trainer, its captured hipGraph and its dozen HIP streams - the same steps measured
    8 - 10 % slower than alone (profiles/round5_secondary_ab.log: ResNet-50 42.0 vs 38.8 ms, 1024x320 34.9 vs 31.8 ms on one box, which
    is the whole of the "regression" VERDICT round 4 saw between round 3's stand-alone numbers and round 4's in-process ones).
    A failing entry reports its error instead of hiding the others."""
    out = {}
    for name in OTHER_CONFIGS:
        t0 = time.perf_counter()
        cmd = [sys.executable, os.path.abspath(__file__), "--_other", name, "--other_steps", str(args.other_steps)]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError("child exited with %d: %s" % (r.returncode, r.stderr[-600:]))
            out[name] = json.loads(lines[-1])
        except Exception as e:                          # one workload failing must not hide the others
            out[name] = {"error": repr(e)}
        out[name]["wall_s"] = time.perf_counter() - t0
        print("[bench] other_configs.%s: %s" % (name, {k: v for k, v in out[name].items() if k != "workload"}), file=sys.stderr, flush=True)
    return out


def run_other_config(args):
    """Child mode (``--_other NAME``): one `other_configs` entry in this fresh process - same launch path as the headline (eager, four
    HIP streams), synthetic scene batches, fp32; 8 untimed steps (the first steps of a process run 5 - 10 % slower than its steady
    state), then 3 windows of --other_steps steps, median reported.  Prints ONE JSON line."""
    import contextlib
    import tempfile
    from fusiondepth_amd import synthetic
    from fusiondepth_amd.options import MonodepthOptions
    from fusiondepth_amd.trainer import Trainer
    torch.manual_seed(20260929)
    n = max(args.other_steps, 1)
    WARM = 8

    def cleanup():
        import gc
        gc.collect()
        torch.cuda.empty_cache()

    def trainer_cfg(layers, H, W, bs):
        opt = MonodepthOptions().parse(["--num_layers", str(layers), "--weights_init", "scratch", "--batch_size", str(bs), "--height", str(H),
                                        "--width", str(W)])
        with contextlib.redirect_stdout(sys.stderr):
            tr = Trainer(opt, verbose=False)
        pool = []
        for i in range(3):
            mbs = [synthetic.make_scene_batch(tr.batch_size, H, W, seed=4321 + 17 * i + j, clutter=0.5) for j in range(tr.accumulate_step)]
            for mb in mbs:
                mb.pop("depth_gt", None)
                for f in (-1, 1):
                    mb.pop(("T_gt", f), None)
            pool.append(tr.stack_micro_batches(mbs))
        k = [0]

        def step():
            k[0] += 1
            return tr.train_step(pool[k[0] % len(pool)])
        st, losses = _short_run(step, n, warm=WARM)
        dt = st["dt"]
        loss = float(losses["loss"].detach())
        r = _timing_fields(st, opt.batch_size)
        r.update({"final_loss": loss if loss == loss else None,
             "final_loss_photometric": float(sum(losses["loss/%d" % s_].detach() for s_ in range(4)) / 4.0),
             "params_finite": bool(torch.isfinite(tr.flat.flat_param).all()),
             "workload": "ResNet-%d, %dx%d, --batch_size %d (= %d micro-batches of %d stacked), fwd+bwd+Adam" % (layers, W, H, opt.batch_size, tr.accumulate_step, tr.batch_size),
             "arith": ARITH_NOTE})
        key = (layers, H, W)
        if key in CONV_GFLOP_FWD_BWD:
            r["step_mfma_frac"] = CONV_GFLOP_FWD_BWD[key] * 1e9 * opt.batch_size / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS
        return r

    def refiner_cfg():
        from fusiondepth_amd.refiner import Refiner
        base = ["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12", "--height", "192", "--width", "640"]
        folder = tempfile.mkdtemp(prefix="fd_stage1_")
        with contextlib.redirect_stdout(sys.stderr):
            tr = Trainer(MonodepthOptions().parse(base + ["--log_dir", folder, "--model_name", "stage1"]), verbose=False)
            tr.save_model("stage1")
            w = os.path.join(tr.log_path, "models", "weights_stage1")
            del tr
            cleanup()
            rf = Refiner(MonodepthOptions().parse(base + ["--refine_load_weights_folder", w]), verbose=False)
        B = rf.batch_size
        gen = torch.Generator(device="cuda"); gen.manual_seed(5)
        pool = []
        for i in range(3):
            inp = synthetic.make_batch(B, 192, 640, seed=77 + i)
            inp["inf_gdc"] = torch.empty(B, 1, 192, 640, device="cuda").uniform_(0.05, 1.5, generator=gen)
            pool.append(inp)
        k = [0]

        def step():
            # a loader that reads one batch ahead (Refiner.run_epoch): the next batch's frozen forward passes overlap this step; every
            # step still runs one frozen block and one refine-decoder forward / backward / Adam
            k[0] += 1
            return rf.train_step(pool[k[0] % 3], pool[(k[0] + 1) % 3])
        st, losses = _short_run(step, n, warm=WARM)
        with _flop_tally() as tally:       # one step of the loop above: the announced batch's frozen block + this batch's refine-decoder passes
            step()
        loss = float(losses["loss"].detach())
        r = _timing_fields(st, B)
        tf = tally.flops() / st["dt"] / 1e12
        r.update({"final_loss": loss if loss == loss else None, "step_conv_tflops": tf, "step_mfma_frac": tf / PEAK_FP32_MFMA_TFLOPS,
                  "step_mfma_frac_note": "direct-equivalent flops of every convolution call of one step (counted by shape: frozen encoders "
                                         "forward only, refine decoder forward + both gradients) over the fp32 MFMA peak",
                "workload": "Refiner.train_step (refiner.py:272-278), 640x192, --batch_size 12 = one optimiser step per batch of %d, stage-1 "
                            "networks (ResNet-18) frozen, refine2d decoder trained; three batches in turn, the next one announced to train_step "
                            "(its frozen block is issued ahead, tuning.host.refiner_prefetch)" % B})
        return r

    def completor_cfg():
        from fusiondepth_amd.completor import Completor
        o = MonodepthOptions().parse(["--weights_init", "scratch", "--batch_size", "12", "--completion_num_layers", "18"])
        with contextlib.redirect_stdout(sys.stderr):
            cp = Completor(o, verbose=False)
        H, W = o.height, o.width
        mbs = [synthetic.make_batch(cp.batch_size, H, W, seed=31 + i) for i in range(cp.accumulate_step)]
        inp = cp.stack_micro_batches(mbs) if cp.stack_microbatches else mbs
        st, losses = _short_run(lambda: cp.train_step(inp), n, warm=WARM)
        with _flop_tally() as tally:
            cp.train_step(inp)
        loss = float(losses["loss"].detach())
        imgs = cp.batch_size * cp.accumulate_step
        r = _timing_fields(st, imgs)
        tf = tally.flops() / st["dt"] / 1e12
        r.update({"final_loss": loss if loss == loss else None, "step_conv_tflops": tf, "step_mfma_frac": tf / PEAK_FP32_MFMA_TFLOPS,
                  "step_mfma_frac_note": "direct-equivalent flops of every convolution call of one step (counted by shape) over the fp32 MFMA peak",
                "workload": "Completor.train_step (completor.py), %dx%d, --batch_size 12 = %d micro-batch(es) of %d, --completion_num_layers 18 "
                            "(the configuration of rounds 2-3's scripts/bench_config5.py)" % (W, H, cp.accumulate_step, cp.batch_size)})
        return r

    fns = {"r50_640x192_b8": lambda: trainer_cfg(50, 192, 640, 8), "r18_1024x320_b8": lambda: trainer_cfg(18, 320, 1024, 8),
           "refiner_640x192": refiner_cfg, "completor_1216x352": completor_cfg}
    res = fns[args._other]()
    res["process"] = "own process (bench.py --_other %s)" % args._other
    print(json.dumps(res), flush=True)


def dp_probe(tr, step_fn, t_step, barrier, reps=3):
    """Multi-rank only, after the timed region: how many ranks answered, what the gradient exchange costs on its own and how
    much of it the backward pass hides.  exposed = (step with exchange) - (step without); overlap_frac = 1 - exposed / allreduce."""
    world = dist.get_world_size()
    n_over = tr.grad_sync.n_overlapped                      # of the last timed step
    ones = torch.ones(1, device="cuda")
    dist.all_reduce(ones)
    g = tr.flat.flat_grad
    barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    dist.all_reduce(g)                                     # warm
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        dist.all_reduce(g)
    ev[1].record()
    torch.cuda.synchronize()
    ar_ms = ev[0].elapsed_time(ev[1]) / reps
    tr.flat.zero_grad()
    tr.grad_sync.skip_comm = True                          # replicas drift apart from here on: this is the last thing the run does
    barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        step_fn()
    barrier()
    t_nocomm = (time.perf_counter() - t0) / reps
    tr.grad_sync.skip_comm = False
    tt = torch.tensor([ar_ms, t_nocomm], device="cuda", dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ar_ms, t_nocomm = float(tt[0]), float(tt[1])
    exposed_ms = max(0.0, 1e3 * (t_step - t_nocomm))
    return {"ranks_seen": int(ones.item()), "backend": dist.get_backend(), "allreduce_ms": ar_ms,
            "allreduce_bytes": int(g.numel()) * 4, "allreduce_buckets": len(tr.grad_sync.buckets),
            "buckets_overlapped": n_over, "ms_per_step_no_exchange": 1e3 * t_nocomm,
            "exposed_exchange_ms": exposed_ms, "overlap_frac": max(0.0, min(1.0, 1.0 - exposed_ms / ar_ms)) if ar_ms > 0 else None}


def self_launch(args):
    """Plain ``python bench.py --gpus N`` (no torchrun environment): start N ranks of this script through torch.distributed.run."""
    n_dev = torch.cuda.device_count()
    env = dict(os.environ)
    if n_dev < args.gpus:
        if not args.share_device:
            sys.exit("bench.py --gpus %d: only %d device(s) visible (torch.cuda.device_count()); refusing to oversubscribe "
                     "(--share_device runs the ranks over gloo on the visible devices: a functional check, not a measurement)"
                     % (args.gpus, n_dev))
        env["FD_DIST_BACKEND"] = "gloo"
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a torchrun environment: launching %s" % (args.gpus, " ".join(cmd[1:9])), file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args._other:
        return run_other_config(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    from fusiondepth_amd import dp, synthetic
    from fusiondepth_amd.options import MonodepthOptions
    from fusiondepth_amd.trainer import Trainer
    rank, world, local_rank = dp.init_from_env()
    torch.manual_seed(20260928 + rank)          # weights (CPU generator) and the loss path's tie-break noise (device generator): a run is reproducible
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d: launch it with --nproc-per-node %d (or plainly, it launches "
                 "itself)" % (args.gpus, world, args.gpus))
    n_dev = torch.cuda.device_count()
    if world > n_dev and not args.share_device:
        sys.exit("bench.py: %d ranks but only %d device(s) visible" % (world, n_dev))
    torch.cuda.set_device(local_rank % n_dev)
    opt = MonodepthOptions().parse(["--num_layers", str(args.num_layers), "--weights_init", "scratch", "--batch_size",
                                    str(args.batch_size), "--height", str(args.height), "--width", str(args.width)])
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):         # the trainer's banner: stdout carries the ONE JSON line and nothing else
        tr = Trainer(opt, rank=rank, world_size=world, verbose=(rank == 0))
    tr.stack_microbatches = not args.no_stack
    # A pool of distinct step batches, resident in HBM.  The reference's loader yields the step's 12 images as 2 micro-batches of 6;
    # the stacked step consumes them as one batch: concatenate once, outside the timed region.
    n_pool = args.pool if args.pool > 0 else min(48, args.steps + max(args.warmup, 2) + 4)
    if args.probe_only:
        n_pool = 1                           # the probes need one batch (and the profile of --probe_only stays free of generator kernels)
    pool = []
    for i in range(n_pool):
        mbs_i = [synthetic.make_scene_batch(tr.batch_size, args.height, args.width, seed=1234 + 1009 * rank + 17 * i + j, clutter=0.5)
                 for j in range(tr.accumulate_step)]
        for mb in mbs_i:
            mb.pop("depth_gt", None)                 # training batches: the validation batch below keeps its ground truth
            for f in (-1, 1):
                mb.pop(("T_gt", f), None)
        pool.append((mbs_i, tr.stack_micro_batches(mbs_i) if tr.stack_microbatches else mbs_i))
    val_batch = synthetic.make_scene_batch(tr.batch_size, args.height, args.width, seed=99991 + rank)
    mbs, eager_in = pool[0]
    cursor = [0]

    def next_batch():
        cursor[0] += 1
        return pool[cursor[0] % n_pool]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.probe_only:
        if rank == 0:
            print(json.dumps(roofline_probes(args, tr, eager_in if tr.stack_microbatches else mbs[0])))
        return

    def eager_step():
        return tr.train_step(next_batch()[1])

    def graph_step():
        return tr.train_step_graphed(next_batch()[0])

    def timed(fn, n):
        barrier()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        barrier()
        return (time.perf_counter() - t) / n

    barrier(); barrier()          # the first collective builds the communicator (seconds): keep it out of every measurement
    abs_rel_before = float(tr.val_metrics([val_batch])["de/abs_rel"])
    # Launch path: eager issue on the four module streams.  Rounds 1-4 timed a hipGraph replay of the step against it during the
    # warm-up and took the faster one; the replay has lost every such trial since round 1 (round 5, profiles/round5_eager_vs_trial.log:
    # 22.4 vs 19.1 ms per step - hipGraphLaunch costs the host more per kernel node than the eager launch it replaces, and its
    # executor overlaps the branches less than real streams do), and the trial itself changes nothing for the eager path
    # (19.03 - 19.22 ms with it, 18.97 - 19.20 without), so it is gone: --graph still forces the replay path.
    launch = "graph" if args.graph else "eager"
    if launch == "graph" and world > 1:
        launch = "eager"           # a capture that fails on one rank only would leave the others waiting in the next collective
    step_fn = eager_step if launch == "eager" else graph_step
    # six extra untimed steps of the chosen path, so that the caching allocator, the weight-layout caches, the batched re-layout
    # plan (built after the first step) and the clocks are in their steady state before the W warm-up steps even when W is 0 or 1
    # (scripts/secondary_ab.py prints the first eight steps of a process one by one: the first costs 2x, the next ~seven 5 - 10 % more
    # than the steady state)
    for _ in range(6):
        step_fn()
    for _ in range(max(args.warmup, 0 if launch == "eager" else 2)):     # graph mode: 1 eager warm-up + 1 capture/replay
        step_fn()
    # K timed steps, bracketed by barrier + synchronize on both sides - `--windows` times over (default 3): a window of 20 steps is
    # 0.4 s, short enough for clock ramps and neighbours on the box to move it by a few percent, so the line reports the MEDIAN
    # window (value, ms_per_step) with the fastest / slowest beside it.  Every window is the MAX over ranks.
    win, win_host, win_ranks = [], [], []
    for _ in range(max(args.windows, 1)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            losses = step_fn()
        t_host = time.perf_counter() - t0            # host time to ISSUE the steps (no sync): == dt when launch-bound
        per_rank = None
        if world > 1:
            torch.cuda.synchronize()
            t_own = time.perf_counter() - t0         # this rank's own K steps (its last all-reduce included), before it waits for the others
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            mine = torch.tensor([t_own, t_host], device="cuda", dtype=torch.float64)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [(float(a[0]), float(a[1])) for a in allr]
            t_host = max(h for _, h in per_rank)      # the slowest host decides when the step's last launch is out
        win.append(dt); win_host.append(t_host); win_ranks.append(per_rank)
    order = sorted(range(len(win)), key=lambda i: win[i])
    mid = order[len(order) // 2]                 # the median window (an even count: the slower of the two middle ones)
    dt, t_host = win[mid], win_host[mid]
    images = args.steps * opt.batch_size * world
    loss_val = float(losses["loss"].detach())
    # The SI-log term of a scale is NaN by definition when no LiDAR return passes its validity mask (mean / variance of an
    # empty set - trainer.py:577-589 behaves the same, tests/test_gpu_losspath.py::test_empty_lidar_mask_*).  The scene feed keeps
    # the returns consistent with the scene the networks learn, so the mask stays populated; a NaN here fails the run.
    photo_val = float(sum(losses["loss/%d" % s_].detach() for s_ in range(4) if ("loss/%d" % s_) in losses) / 4.0)
    params_finite = bool(torch.isfinite(tr.flat.flat_param).all())
    n_steps_run = tr.adam_step_count
    flat64 = tr.flat.flat_param.double()
    param_checksum = [float(flat64.sum()), float(flat64.abs().sum())]        # equal between two runs of one build: the step is deterministic
    abs_rel_after = float(tr.val_metrics([val_batch])["de/abs_rel"])
    rank_spread = None
    if win_ranks[mid]:
        own = [1e3 * a / args.steps for a, _ in win_ranks[mid]]
        rank_spread = {"ms_per_step_by_rank": own, "ms_per_step_min": min(own), "ms_per_step_max": max(own),
                       "host_issue_ms_per_step_by_rank": [1e3 * h / args.steps for _, h in win_ranks[mid]],
                       "note": "each rank's own time for the window's steps up to its local synchronize (before the closing barrier): "
                               "the spread is what the stragglers cost; ranks are pinned to their GPU's NUMA node (dp.pin_to_gpu_numa)"}
    if rank == 0:
        print("[bench] timed %d steps in %.3f s (host issue time %.3f s; median of %d windows: %s ms/step)"
              % (args.steps, dt, t_host, len(win), ", ".join("%.2f" % (1e3 * w / args.steps) for w in win)), file=sys.stderr, flush=True)
    result = {
        "metric": "training images/sec (%dx%d, ResNet-%d, 4-beam)" % (args.width, args.height, args.num_layers), "value": images / dt, "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "arith": ARITH_NOTE, "data": "synthetic",
        "windows": {"n": len(win), "steps_each": args.steps, "ms_per_step": [1e3 * w / args.steps for w in win],
                    "ms_per_step_min": 1e3 * min(win) / args.steps, "ms_per_step_max": 1e3 * max(win) / args.steps,
                    "value_min": images / max(win), "value_max": images / min(win), "reported": "median window",
                    "host_issue_ms_per_step": 1e3 * t_host / args.steps,
                    "host_issue_note": "host time to issue one step's launches (no synchronisation); N > 1: the MAX over ranks"},
        "config": {"workload": "ResNet-%d encoders + DepthDecoder + PoseDecoder, %dx%d, 4-beam LiDAR, --batch_size %d per GPU "
                               "(= %d accumulated micro-batches of %d), frames [0,-1,1], 4 scales, fwd+bwd+Adam"
                               % (args.num_layers, args.width, args.height, opt.batch_size, tr.accumulate_step, tr.batch_size),
                   "global_batch": opt.batch_size * world, "parallelism": "dp%d" % world, "launch": "eager (4 HIP streams)" if launch == "eager" else "hipGraph replay",
                   "micro_batches": "stacked (grouped BatchNorm)" if tr.stack_microbatches else "sequential"},
        "final_loss": loss_val if loss_val == loss_val else None, "final_loss_photometric": photo_val,
        "params_finite": params_finite, "optimizer_steps_run": n_steps_run, "distinct_step_batches": n_pool, "param_checksum": param_checksum,
        "abs_rel_heldout_scene": {"before": abs_rel_before, "after": abs_rel_after},
    }
    key = (args.num_layers, args.height, args.width)
    if key in CONV_GFLOP_FWD_BWD:
        tf = CONV_GFLOP_FWD_BWD[key] * 1e9 * (images / world) / dt / 1e12
        result["step_mfma_frac"] = tf / PEAK_FP32_MFMA_TFLOPS
        result["step_conv_tflops_per_gpu"] = tf
    if world > 1:
        result["rank_spread"] = rank_spread
        result.update(dp_probe(tr, step_fn, dt / args.steps, barrier))
        if result["ranks_seen"] != args.gpus or (result["backend"] != "nccl" and not args.share_device):
            sys.exit("bench.py: %d ranks answered over %r, expected %d over nccl (= RCCL)" % (result["ranks_seen"], result["backend"], args.gpus))
    if rank == 0 and not args.no_roofline:
        result.update(roofline_probes(args, tr, eager_in if tr.stack_microbatches else mbs[0]))
        print("[bench] roofline probes done", file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and not args.no_other_configs and (args.num_layers, args.height, args.width, args.batch_size) == (18, 192, 640, 12):
        del tr, pool, eager_in, mbs, val_batch, step_fn
        import gc
        gc.collect(); torch.cuda.empty_cache()
        result["other_configs"] = other_configs(args)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # rank 0 alone (N > 1: after the process group is gone - the other ranks have left, no collective waits on the CPU run)
    if rank == 0 and not args.no_cpu_baseline:
        result["cpu_baseline_1thread"], result["cpu_baseline"] = cpu_baseline(args)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if not (loss_val == loss_val and params_finite):
        sys.exit(3)            # a degenerate run is not a measurement


if __name__ == "__main__":
    main()
