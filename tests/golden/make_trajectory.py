"""Generator of tests/golden/trajectory_r18_128x192_b2.npz - SURVEY.md section 7 step 0: a 20-optimiser-step trajectory of the
training step (trainer.py:237-248, 268-319, 425-596 + Adam), produced by the float32 CPU oracle (the loss path of the oracle is
pinned against the imported reference, tests/test_oracle_golden.py; the ResNet trunk is the oracle's restatement of torchvision).

    python tests/golden/make_trajectory.py            # ~2 minutes on 8 cores; inputs are regenerated from seeds by the test

Stored per step: every entry of the loss dict.  Stored once: the disparity at scale 0 after the last step on a held-out batch (strided)
and the parameter movement statistics.  Two trajectories from the same initial state: float32 and float64 (the float64 one
shows how far apart two exact-arithmetic-equivalent evaluations drift through Adam, i.e. what tolerance the comparison can carry)."""
import copy
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

H, W, B, STEPS, SEED = 128, 192, 2, 20, 3


def scaled_roi(H, W):
    return (max(int(round(76 * H / 192)), 2), min(int(round(190 * H / 192)), H - 2), 2, W - 2)


def batch(seed):
    """Same construction as tests/test_gpu_trainer.py::_batch."""
    inp, rng = gin.batch_inputs(seed, B, H, W)
    roi = scaled_roi(H, W)
    for i, f in enumerate((0, -1, 1)):
        beam = gin.lidar_4beam(np.random.RandomState(seed + 10 + i), B, H, W)
        beam = np.where(beam > 0, 0.035 + (beam - 0.05) * (0.035 / 0.6), 0).astype(np.float32)
        two = np.stack([np.stack(OS.scatter_2channel_c(beam[b, 0], roi)) for b in range(B)])
        inp[("2channel", f, 0)] = torch.from_numpy(two)
        if f == 0:
            inp["2channel"] = torch.from_numpy(two)
            inp["4beam"] = torch.from_numpy(beam)
    noise = [torch.from_numpy(np.random.RandomState(seed + 50 + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
    return inp, noise


def models(opt):
    m = OT.build_models(opt, SEED)
    for k, net in m.items():
        gin.fill_params(net, 100 + len(k))
    return m


def run(dtype):
    opt = OT.default_opt(height=H, width=W, batch_size=B, num_layers=18, learning_rate=1e-4)
    m = models(opt)
    if dtype == torch.float64:
        m = {k: net.double() for k, net in m.items()}
    ot = OT.OracleTrainer(opt, models=m)
    rec = {}
    for step in range(STEPS):
        inp, noise = batch(900 + step)
        if dtype == torch.float64:
            inp = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
            noise = [n.double() for n in noise]
        _, losses = ot.micro_step(inp, noise)
        for k, v in losses.items():
            rec.setdefault(k, []).append(float(v))
        print(dtype, step, rec["loss"][-1], flush=True)
    return rec, ot


if __name__ == "__main__":
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    out = {}
    for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
        rec, ot = run(dt)
        for k, v in rec.items():
            out["%s/%s" % (tag, k.replace("/", "_"))] = np.asarray(v, np.float64)
        p = torch.cat([q.detach().reshape(-1).double() for q in OT.trainable_parameters(ot.models)])
        out["%s/param_sum" % tag] = np.float64(p.sum())
        out["%s/param_l2" % tag] = np.float64(p.norm())
    np.savez_compressed(os.path.join(HERE, "trajectory_r18_128x192_b2.npz"), **out)
    print("drift f32 vs f64 per step:", np.abs(out["f32/loss"] - out["f64/loss"]) / out["f64/loss"])
