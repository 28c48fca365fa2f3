"""Refiner step (bench.py's config-5 workload) with the frozen networks replayed (fd_replay) vs issued eagerly, in one process:
host issue time and step time of both, state of every Replayable.   python scripts/refiner_replay_probe.py"""
import contextlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic, tuning
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
from fusiondepth_amd.refiner import Refiner

base = ["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12", "--height", "192", "--width", "640"]
folder = tempfile.mkdtemp(prefix="fd_stage1_")
with contextlib.redirect_stdout(sys.stderr):
    tr = Trainer(MonodepthOptions().parse(base + ["--log_dir", folder, "--model_name", "stage1"]), verbose=False)
    tr.save_model("stage1")
    w = os.path.join(tr.log_path, "models", "weights_stage1")
    del tr
    rf = Refiner(MonodepthOptions().parse(base + ["--refine_load_weights_folder", w]), verbose=False)
B = rf.batch_size
inp = synthetic.make_batch(B, 192, 640, seed=77)
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
inp["inf_gdc"] = torch.empty(B, 1, 192, 640, device="cuda").uniform_(0.05, 1.5, generator=gen)


def run(n=20):
    for _ in range(6):
        rf.train_step(inp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        rf.train_step(inp)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * th / n, 1e3 * (time.perf_counter() - t0) / n


for rnd in range(3):
    for mode in (False, True):
        tuning.host.replay_frozen = mode
        h, s = run()
        print("replay_frozen=%-5s host issue %.2f ms / step, step %.2f ms = %.1f images/s" % (mode, h, s, B / s * 1e3), flush=True)
print({k: (r.disabled, len(r.plans), [p[0].n_recs for p in r.plans.values()]) for k, r in rf._replays.items()})
