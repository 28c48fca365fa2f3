"""Generator of tests/golden/absrel_r18_192x640_b2.npz - the north star's "AbsRel within 0.001 of the reference after equal steps"
(BASELINE.json; SURVEY.md section 8d proxy ii; reference trainer.py:598-630, evaluate_depth.py:42-60) on a scene where AbsRel
means something: ``fusiondepth_amd.synthetic.make_scene_batch`` - a ground-truth depth field, frames -1 / +1 rendered through it
with a ground-truth ego-motion, LiDAR returns sampled from it, ``depth_gt`` = the field at KITTI's 375x1242.

    python tests/golden/make_absrel.py            # ~12 minutes on 8 cores

20 optimiser steps of the CPU oracle trainer (ResNet-18, 192x640, --batch_size 2, the reference's default --learning_rate 1e-4 ->
Adam lr 2.5e-5) from the deterministic initial state of the trainer tests, one fresh scene batch per step; at steps 0, 2, .. 20
the monitoring metrics of
``Trainer.compute_depth_losses`` (bilinear to 375x1242, Garg crop, median scaling, clamp, layers.compute_depth_errors) on two
held-out scene batches in eval mode.  Stored for a float32 AND a float64 run from the same state: the float64 one is the ground
truth, their distance is what the reference's own arithmetic drifts by - from-scratch training with train-mode BatchNorm over two
images and Adam's sign-like first updates is chaotic: the two runs agree to 4e-7 in AbsRel after 2 steps, 2e-4 after 4, 4e-3 after
6 and 1e-2 after 20 (at Adam lr 1e-4 over 50 steps: 9e-2 after 20).  The test (tests/test_gpu_trainer.py)
regenerates the inputs from the seeds below and runs the HIP trainer over the same 20 steps.

How far apart may two CORRECT float32 implementations be?  One float32-vs-float64 pair is one draw of a chaotic process, so the
fixture also holds ENSEMBLE float32 runs of the same oracle whose initial weights were each moved to a neighbouring float32 (one
ulp, random direction): the spread of AbsRel over {float32, float64, one-ulp runs} at a checkpoint is the yardstick the test
uses where it exceeds the north star's 0.001."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import inputs as gin            # noqa: E402
from oracle import scatter as OS, trainer as OT      # noqa: E402

H, W, B, STEPS, EVERY, SEED, LR = 192, 640, 2, 20, 2, 3, 1e-4      # the reference default --learning_rate 1e-4 at --batch_size 2 -> Adam lr 2.5e-5 (trainer.py:38)
TRAIN_SEED, VAL_SEEDS = 2000, (7001, 7002)
ENSEMBLE = 3        # extra float32 runs whose initial weights are moved by ONE float32 ulp each (random direction)
ENSEMBLE_SHORT, SHORT_STEPS = 6, 8     # six more such runs over the first 8 steps only: the checkpoints where 0.001 is still in reach


def scene_batch(seed):
    """(inputs, tie-break noise) on the CPU: the scene generator of the package with the oracle's C scatter."""
    from fusiondepth_amd import functional as FD, synthetic

    def scatter(beam):
        roi = FD.scaled_roi(beam.shape[2], beam.shape[3])
        return torch.from_numpy(np.stack([np.stack(OS.scatter_2channel_c(beam[b, 0].numpy(), roi)) for b in range(beam.shape[0])]))

    inp = synthetic.make_scene_batch(B, H, W, seed=seed, device="cpu", scatter=scatter)
    for f in (-1, 1):
        inp.pop(("T_gt", f))
    noise = [torch.from_numpy(np.random.RandomState(seed + 50 + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
    return inp, noise


def models(opt):
    m = OT.build_models(opt, SEED)
    for k, net in m.items():
        gin.fill_params(net, 100 + len(k))
    return m


def evaluate(ot, dtype):
    """Mean monitoring metrics over the held-out batches, eval-mode BatchNorm, depth networks only (trainer.py:390-409)."""
    from oracle import layers as OL
    import torch.nn.functional as F
    for m in ot.models.values():
        m.eval()
    acc = np.zeros(7)
    with torch.no_grad():
        for seed in VAL_SEEDS:
            inp, _ = scene_batch(seed)
            x, two = inp[("color_aug", 0, 0)].to(dtype), inp["2channel"].to(dtype)
            disp = ot.models["depth"](ot.models["encoder"](x), beam_features=ot.models["beam_encoder"](two))[("disp", 0)]
            depth = OL.disp_to_depth(F.interpolate(disp, [H, W], mode="bilinear", align_corners=False), 0.1, 100.0)[1]
            acc += np.asarray(OT.compute_depth_losses(depth, inp["depth_gt"].to(dtype)), np.float64)
    for m in ot.models.values():
        m.train()
    return acc / len(VAL_SEEDS)


def run(dtype, ulp_seed=None, steps=None):
    opt = OT.default_opt(height=H, width=W, batch_size=B, num_layers=18, learning_rate=LR)
    m = models(opt)
    if ulp_seed is not None:        # the same reference, one rounding away: every weight to a neighbouring float32
        gen = torch.Generator().manual_seed(ulp_seed)
        with torch.no_grad():
            for net in m.values():
                for prm in net.parameters():
                    up = torch.rand(prm.shape, generator=gen) < 0.5
                    prm.copy_(torch.nextafter(prm, torch.where(up, torch.full_like(prm, float("inf")), torch.full_like(prm, -float("inf")))))
    if dtype == torch.float64:
        m = {k: net.double() for k, net in m.items()}
    ot = OT.OracleTrainer(opt, models=m)
    assert abs(ot.hp.learning_rate - 2.5e-5) < 1e-12 and ot.hp.accumulate_step == 1
    losses, metrics = [], [evaluate(ot, dtype)]
    for step in range(STEPS if steps is None else steps):
        inp, noise = scene_batch(TRAIN_SEED + step)
        if dtype == torch.float64:
            inp = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
            noise = [n.double() for n in noise]
        _, l = ot.micro_step(inp, noise)
        losses.append(float(l["loss"]))
        if (step + 1) % EVERY == 0:
            metrics.append(evaluate(ot, dtype))
            print(dtype, step + 1, "loss %.5f" % losses[-1], "abs_rel %.5f" % metrics[-1][0], flush=True)
    return np.asarray(losses), np.asarray(metrics)


if __name__ == "__main__":
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    path = os.path.join(HERE, "absrel_r18_192x640_b2.npz")
    if "--ensemble-only" in sys.argv or "--short-only" in sys.argv:      # keep the runs the existing fixture already holds
        out = dict(np.load(path))
    else:
        out = {"steps": np.int64(STEPS), "every": np.int64(EVERY)}
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            out["%s/loss" % tag], out["%s/metrics" % tag] = run(dt)
    for k in range(0 if "--short-only" in sys.argv else ENSEMBLE):
        out["f32p%d/loss" % k], out["f32p%d/metrics" % k] = run(torch.float32, ulp_seed=500 + k)
        print("one-ulp run %d |abs_rel - f32|:" % k, np.abs(out["f32p%d/metrics" % k][:, 0] - out["f32/metrics"][:, 0]))
    out["ensemble"] = np.int64(ENSEMBLE)
    if "--short-only" in sys.argv or "--ensemble-only" not in sys.argv:
        for k in range(ENSEMBLE_SHORT):
            out["f32s%d/loss" % k], out["f32s%d/metrics" % k] = run(torch.float32, ulp_seed=900 + k, steps=SHORT_STEPS)
            print("short one-ulp run %d |abs_rel - f32|:" % k, np.abs(out["f32s%d/metrics" % k][:, 0] - out["f32/metrics"][:len(out["f32s%d/metrics" % k]), 0]))
        out["ensemble_short"] = np.int64(ENSEMBLE_SHORT)
    np.savez_compressed(path, **out)
    print("abs_rel f32:", out["f32/metrics"][:, 0])
    print("abs_rel f64:", out["f64/metrics"][:, 0])
    print("|f32 - f64|:", np.abs(out["f32/metrics"][:, 0] - out["f64/metrics"][:, 0]))
