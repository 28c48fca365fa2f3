"""Where the host time of an eager training step goes: cProfile over 20 steady-state steps, functions by own time.
usage: host_profile.py [n_rows]"""
import cProfile, io, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
opt = MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12", "--height", "192", "--width", "640"])
tr = Trainer(opt, verbose=False)
pool = []
for i in range(6):
    mbs = [synthetic.make_scene_batch(tr.batch_size, 192, 640, seed=1234 + 17 * i + j, clutter=0.5) for j in range(tr.accumulate_step)]
    for mb in mbs:
        mb.pop("depth_gt", None)
        for f in (-1, 1):
            mb.pop(("T_gt", f), None)
    pool.append(tr.stack_micro_batches(mbs))
for i in range(8):
    tr.train_step(pool[i % 6])
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for i in range(20):
    tr.train_step(pool[i % 6])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("un-profiled: issue %.2f ms / step, wall %.2f ms / step" % (t_issue * 50, t_all * 50))
# issue time of ONE step into empty queues (no back-pressure from the GPU: the host's own cost)
single = []
for i in range(9):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(pool[i % 6])
    single.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
single.sort()
print("one step into empty queues: issue median %.2f ms (min %.2f, max %.2f)" % (single[4] * 1e3, single[0] * 1e3, single[-1] * 1e3))
pr = cProfile.Profile()
pr.enable()
for i in range(20):
    tr.train_step(pool[i % 6])
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(int(sys.argv[1]) if len(sys.argv) > 1 else 45)
print(s.getvalue())
