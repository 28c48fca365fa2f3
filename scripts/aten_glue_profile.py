import os, sys, collections, traceback
sys.path.insert(0, "/root/repo")
import torch
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
from fusiondepth_amd import synthetic
opt = MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12", "--height", "192", "--width", "640"])
tr = Trainer(opt, rank=0, world_size=1, verbose=False)
mbs = [synthetic.make_batch(tr.batch_size, 192, 640, seed=1234 + i) for i in range(tr.accumulate_step)]
stacked = tr.stack_micro_batches(mbs)
for _ in range(3): tr.train_step(stacked)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    tr.train_step(stacked)
torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::cat", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::mul", "aten::sum", "aten::mean", "aten::randn", "aten::normal_", "aten::div", "aten::to", "aten::_to_copy", "aten::select_backward", "aten::slice_backward"):
        st = [s for s in (ev.stack or []) if "/root/repo/" in s]
        cnt[(ev.name, st[0] if st else "autograd/engine")] += 1
for (n, s), c in cnt.most_common(40):
    print("%4d  %-22s %s" % (c, n, s.replace("/root/repo/", "")))
