import os, sys, tempfile, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
from fusiondepth_amd.refiner import Refiner
base = ["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12", "--height", "192", "--width", "640"]
folder = tempfile.mkdtemp()
tr = Trainer(MonodepthOptions().parse(base + ["--log_dir", folder, "--model_name", "s1"]), verbose=False)
tr.save_model("s1"); w = os.path.join(tr.log_path, "models", "weights_s1"); del tr
rf = Refiner(MonodepthOptions().parse(base + ["--refine_load_weights_folder", w]), verbose=False)
B = rf.batch_size
inp = synthetic.make_batch(B, 192, 640, seed=77)
inp["inf_gdc"] = torch.empty(B, 1, 192, 640, device="cuda").uniform_(0.05, 1.5)
for _ in range(4): rf.train_step(inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): rf.train_step(inp)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) * 100, (t2 - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(5): rf.train_step(inp)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
