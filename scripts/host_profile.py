"""Where does the HOST spend the issue time of a step?  cProfile over 5 eager optimiser steps of the bench configuration
(the GPU runs behind; one synchronize at the end).   (run on the GPU box)"""
import os, sys, cProfile, pstats, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer

opt = MonodepthOptions().parse(["--batch_size", "12", "--height", "192", "--width", "640", "--weights_init", "scratch"])
tr = Trainer(opt, verbose=False)
mbs = [synthetic.make_batch(tr.batch_size, 192, 640, seed=1234 + i) for i in range(tr.accumulate_step)]
inp = tr.stack_micro_batches(mbs)
for _ in range(4):
    tr.train_step(inp)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    tr.train_step(inp)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("unprofiled: host issue %.2f ms/step, wall %.2f ms/step" % ((t1 - t0) * 200, (t2 - t0) * 200))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.train_step(inp)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(35)
