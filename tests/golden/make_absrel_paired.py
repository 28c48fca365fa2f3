"""Generator of tests/golden/absrel_paired_r18_96x320_b2.npz - the PAIRED form of the north star's "AbsRel within 0.001 of the
reference after equal steps" (reference trainer.py:598-630, layers.py:284-302).

    python tests/golden/make_absrel_paired.py [threads]          # ~1 h on 8 cores (24 oracle runs of 60 optimiser steps)

make_absrel_stat.py compared 6 HIP runs with 6 oracle runs as two UNPAIRED samples: the run-to-run spread of AbsRel over different
data streams (0.06 ... 0.15) then sets the resolution of the test, far above 0.001.  Here every data stream k = 0 .. K-1 is run
THREE times from one initial state on exactly the same batches and tie-break noise:
  * the CPU oracle in float32 ("base"),
  * the CPU oracle in float32 with every initial weight moved to a neighbouring float32 (one ulp, random direction: "ulp") - what
    two equally correct float32 implementations of the reference differ by, amplified by the training dynamics, and
  * the HIP trainer (in the test: tests/test_gpu_trainer.py::test_absrel_paired_gap_vs_oracle).
The statistic is the PAIRED difference per stream, HIP - base, whose stream-to-stream component cancels; its mean over the K streams
with the standard error of that mean is what the test bounds and reports, next to the oracle's own paired gap ulp - base.
K = 12 streams, ResNet-18, 96x320, --batch_size 2, the reference's default learning rate, 60 optimiser steps, AbsRel of two
held-out scenes after 0 / 20 / 40 / 60 steps.  Streams 0 .. 5 are the streams of make_absrel_stat.py (same seeds)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import make_absrel_stat as MS   # noqa: E402
from oracle import trainer as OT      # noqa: E402

K = 12
NAME = "absrel_paired_r18_96x320_b2"
ULP_SEED = 7300


def perturb_one_ulp(models, seed):
    """Every weight to a neighbouring float32 (random direction) - as tests/golden/make_absrel.py::run."""
    gen = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for net in models.values():
            for prm in net.parameters():
                up = torch.rand(prm.shape, generator=gen) < 0.5
                prm.copy_(torch.nextafter(prm, torch.where(up, torch.full_like(prm, float("inf")), torch.full_like(prm, -float("inf")))))


def run(k, ulp):
    opt = MS.oracle_opt()
    m = MS.models(opt, k)
    if ulp:
        perturb_one_ulp(m, ULP_SEED + k)
    ot = OT.OracleTrainer(opt, models=m)
    assert abs(ot.hp.learning_rate - 2.5e-5) < 1e-12 and ot.hp.accumulate_step == 1
    metrics = [MS.evaluate(ot)]
    for step in range(MS.STEPS):
        inp, noise = MS.scene_batch(MS.train_seed(k, step))
        ot.micro_step(inp, noise)
        if (step + 1) in MS.CHECK:
            metrics.append(MS.evaluate(ot))
    print("stream %2d %s abs_rel %s" % (k, "ulp " if ulp else "base", np.asarray(metrics)[:, 0]), flush=True)
    return np.asarray(metrics)


if __name__ == "__main__":
    torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else min(16, max(1, os.cpu_count() or 1)))
    base = np.stack([run(k, False) for k in range(K)])           # [K, len(CHECK), 7]
    ulp = np.stack([run(k, True) for k in range(K)])
    np.savez_compressed(os.path.join(HERE, NAME + ".npz"), base=base, ulp=ulp, check=np.asarray(MS.CHECK, np.int64), streams=np.int64(K),
                        steps=np.int64(MS.STEPS))
    d = ulp[:, :, 0] - base[:, :, 0]
    print("oracle paired gap ulp - base: mean %s, SE %s" % (d.mean(0), d.std(0, ddof=1) / np.sqrt(K)))
