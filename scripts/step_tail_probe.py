import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fusiondepth_amd import synthetic, functional as FD
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
opt = MonodepthOptions().parse(["--num_layers", "18", "--batch_size", "12", "--height", "192", "--width", "640", "--weights_init", "scratch"])
tr = Trainer(opt, verbose=False)
mbs = [synthetic.make_batch(tr.batch_size, 192, 640, seed=1234 + i) for i in range(tr.accumulate_step)]
inp = tr.stack_micro_batches(mbs)
def run(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): tr.train_step(inp)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
for _ in range(6): tr.train_step(inp)
a = run(20)
orig = tr.optimizer_step
def no_opt(scale=1.0):
    tr.adam_step_count += 1
tr.optimizer_step = no_opt
for _ in range(3): tr.train_step(inp)
b = run(20)
tr.optimizer_step = orig
for _ in range(3): tr.train_step(inp)
c = run(20)
print("full step %.2f ms | without Adam + zero_grad + re-layout %.2f ms | full again %.2f ms" % (a, b, c))
