"""Generator of tests/golden/absrel_stat_r18_96x320_b2.npz - the STATISTICAL form of the north star's "AbsRel within 0.001 of the
reference after equal steps" (reference trainer.py:598-630, layers.py:284-302).

    python tests/golden/make_absrel_stat.py            # ~17 minutes on 8 cores

tests/golden/make_absrel.py showed that from-scratch training is chaotic: one float32 trajectory of the reference's own arithmetic
leaves another (same code, weights one ulp apart) by more than 0.001 in AbsRel after ~4 optimiser steps.  Trajectory-by-trajectory
agreement is therefore not a property a correct implementation can have beyond the first steps; what CAN be tested over a long
run is that the HIP trainer and the reference draw from the same DISTRIBUTION.  This fixture holds K = 6 independent runs of the CPU
oracle trainer (ResNet-18, 96x320, --batch_size 2, the reference's default learning rate; the runs share the initial weights and
have each their own stream of scene batches), 60 optimiser steps each, AbsRel of two held-out scenes
(``fusiondepth_amd.synthetic.make_scene_batch``: consistent depth field, frames rendered through it, LiDAR from it, ``depth_gt`` at
375x1242) after 0, 20, 40 and 60 steps.  (Where the untrained network's depths lie outside the 2 m window around every LiDAR
return the SI-log term of that scale is NaN with finite gradients, as in the reference - trainer.py:577-589; the recorded loss is
then NaN, the training goes on; the statistic compared is AbsRel.)  The test (tests/test_gpu_trainer.py::test_absrel_distribution_matches_the_oracle) runs
the HIP trainer from the same K initial states over the same batches and compares means and spreads."""
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

H, W, B, STEPS, LR, K = 96, 320, 2, 60, 1e-4, 6
CHECK = (0, 20, 40, 60)
TRAIN_SEED, VAL_SEEDS = 52000, (57001, 57002)
NAME = "absrel_stat_r18_96x320_b2"


def scene_batch(seed):
    from fusiondepth_amd import functional as FD, synthetic

    def scatter(beam):
        roi = FD.scaled_roi(beam.shape[2], beam.shape[3])
        return torch.from_numpy(np.stack([np.stack(OS.scatter_2channel_c(beam[b, 0].numpy(), roi)) for b in range(beam.shape[0])]))

    inp = synthetic.make_scene_batch(B, H, W, seed=seed, device="cpu", scatter=scatter)
    for f in (-1, 1):
        inp.pop(("T_gt", f))
    noise = [torch.from_numpy(np.random.RandomState(seed + 50 + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
    return inp, noise


def train_seed(k, step):
    return TRAIN_SEED + 1000 * k + step


def oracle_opt():
    return OT.default_opt(height=H, width=W, batch_size=B, num_layers=18, learning_rate=LR)


def models(opt, k):
    """Initial state of every run (the HIP trainer of the test copies it): the deterministic state of the other trainer tests.  The
    runs differ by their stream of scene batches (``train_seed``): from-scratch training with two-image BatchNorm amplifies any
    difference within a few steps (make_absrel.py), so different data order is all it takes to make them independent draws."""
    m = OT.build_models(opt, 3)
    for name, net in m.items():
        gin.fill_params(net, 100 + len(name))
    return m


def evaluate(ot):
    from oracle import layers as OL
    import torch.nn.functional as F
    for m in ot.models.values():
        m.eval()
    acc = np.zeros(7)
    with torch.no_grad():
        for seed in VAL_SEEDS:
            inp, _ = scene_batch(seed)
            disp = ot.models["depth"](ot.models["encoder"](inp[("color_aug", 0, 0)]), beam_features=ot.models["beam_encoder"](inp["2channel"]))[("disp", 0)]
            depth = OL.disp_to_depth(F.interpolate(disp, [H, W], mode="bilinear", align_corners=False), 0.1, 100.0)[1]
            acc += np.asarray(OT.compute_depth_losses(depth, inp["depth_gt"]), np.float64)
    for m in ot.models.values():
        m.train()
    return acc / len(VAL_SEEDS)


def run(k):
    opt = oracle_opt()
    ot = OT.OracleTrainer(opt, models=models(opt, k))
    assert abs(ot.hp.learning_rate - 2.5e-5) < 1e-12 and ot.hp.accumulate_step == 1
    losses, metrics = [], [evaluate(ot)]
    for step in range(STEPS):
        inp, noise = scene_batch(train_seed(k, step))
        losses.append(float(ot.micro_step(inp, noise)[1]["loss"]))
        if (step + 1) in CHECK:
            metrics.append(evaluate(ot))
            print("run %d step %2d loss %.5f abs_rel %.5f" % (k, step + 1, losses[-1], metrics[-1][0]), flush=True)
    return np.asarray(losses), np.asarray(metrics)


if __name__ == "__main__":
    torch.set_num_threads(min(16, max(1, os.cpu_count() or 1)))
    out = {"steps": np.int64(STEPS), "check": np.asarray(CHECK, np.int64), "runs": np.int64(K)}
    L, M = [], []
    for k in range(K):
        l, m = run(k)
        L.append(l); M.append(m)
    out["loss"] = np.stack(L); out["metrics"] = np.stack(M)            # [K, STEPS], [K, len(CHECK), 7]
    np.savez_compressed(os.path.join(HERE, NAME + ".npz"), **out)
    a = out["metrics"][:, :, 0]
    print("abs_rel per run and checkpoint:\n", a)
    print("mean", a.mean(0), "std", a.std(0, ddof=1))
