"""Refiner step: host issue time (no synchronisation inside the window) next to the synchronised step time."""
import contextlib, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic
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
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
pool = []
for i in range(3):
    inp = synthetic.make_batch(B, 192, 640, seed=77 + i)
    inp["inf_gdc"] = torch.empty(B, 1, 192, 640, device="cuda").uniform_(0.05, 1.5, generator=gen)
    pool.append(inp)
for k in range(10):
    rf.train_step(pool[k % 3], pool[(k + 1) % 3])
torch.cuda.synchronize()
N = 30
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(N):
        rf.train_step(pool[k % 3], pool[(k + 1) % 3])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host issue %.2f ms/step, step %.2f ms (%.1f images/s), batch %d" % (1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N, B * N / (t2 - t0), B))
