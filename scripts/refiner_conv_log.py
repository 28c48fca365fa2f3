"""One Refiner step with fd_tuning.log = 1: which kernel family every convolution call takes (stderr lines 'FDCONV ...')."""
import contextlib, os, sys, tempfile
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
tuning.host.replay_frozen = False
for _ in range(2):
    rf.train_step(inp)
torch.cuda.synchronize()
tuning.set_lib(log=1)
rf.train_step(inp)
torch.cuda.synchronize()
tuning.set_lib(log=0)
