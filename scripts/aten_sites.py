"""Which Python lines of the step issue ATen device work (copies, adds, cats, fills ...)?  torch.profiler with stacks over ONE
optimiser step of the bench configuration; prints aten op -> calling source line -> calls per step.   (run on the GPU box)"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer

opt = MonodepthOptions().parse(["--batch_size", "12", "--height", "192", "--width", "640", "--weights_init", "scratch"])
tr = Trainer(opt, verbose=False)
mbs = [synthetic.make_batch(tr.batch_size, 192, 640, seed=1234 + i) for i in range(tr.accumulate_step)]
inp = tr.stack_micro_batches(mbs)
for _ in range(3):
    tr.train_step(inp)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(inp)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    if not any(k.device_time > 0 for k in [ev]) and not ev.kernels:
        continue
    where = "?"
    for fr in ev.stack:
        if "fusiondepth_amd" in fr or "bench.py" in fr:
            where = fr.replace(ROOT + "/", ""); break
    sites[(ev.name, where, len(ev.kernels))] += 1
print("| aten op | kernels | calls | first package frame |\n|---|---:|---:|---|")
for (name, where, nk), n in sorted(sites.items(), key=lambda kv: -kv[1] * max(kv[0][2], 1)):
    if nk:
        print("| %s | %d | %d | %s |" % (name, nk, n, where))
