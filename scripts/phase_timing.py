"""Where a training step's wall time goes (eager launch path, HIP events on the main stream):
   encoders | depth decoder | pose decoder + losses (forward) | backward + Adam.   usage: phase_timing.py [steps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
opt = MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12", "--height", "192", "--width", "640"])
tr = Trainer(opt, rank=0, world_size=1, verbose=False)
mbs = [synthetic.make_scene_batch(tr.batch_size, 192, 640, seed=1234 + i, clutter=0.5) for i in range(tr.accumulate_step)]
batch = tr.stack_micro_batches(mbs)
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
enc, dep, pp, cl = tr.models["encoder"].forward, tr.models["depth"].forward, tr.predict_poses, tr.compute_losses
def w(fn, name):
    def f(*a, **k):
        r = fn(*a, **k); mark(name); return r
    return f
def enc_w(*a, **k):
    r = enc(*a, **k); mark("depth encoder issued+done (main stream)")
    if torch.is_grad_enabled() and r[-1].requires_grad:
        r[-1].register_hook(lambda g: mark("loss bwd + depth decoder bwd (main stream, until d features[-1])"))
    return r
tr.models["encoder"].forward = enc_w
tr.models["depth"].forward = w(dep, "depth decoder fwd")
tr.predict_poses = w(pp, "pose decoder fwd (joins pose encoders)")
tr.compute_losses = w(cl, "warp + losses fwd")
for _ in range(3): tr.train_step(batch)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
acc = {}
for _ in range(n):
    marks.clear(); torch.cuda.synchronize(); mark("start")
    tr.train_step(batch); mark("encoders bwd (4 streams) + Adam + re-layout")
    torch.cuda.synchronize()
    for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
        acc[n1] = acc.get(n1, 0.0) + e0.elapsed_time(e1)
tot = sum(acc.values())
for k, v in acc.items(): print("%-46s %7.2f ms" % (k, v / n))
print("%-46s %7.2f ms" % ("total", tot / n))
