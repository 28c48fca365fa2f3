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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.train_step(inp)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or (ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::")):
        continue
    if ev.name in ("aten::empty", "aten::empty_like", "aten::view", "aten::slice", "aten::select", "aten::as_strided", "aten::detach",
                   "aten::alias", "aten::unsqueeze", "aten::squeeze", "aten::reshape", "aten::expand", "aten::permute", "aten::t",
                   "aten::transpose", "aten::_unsafe_view", "aten::empty_strided", "aten::result_type", "aten::is_nonzero"):
        continue
    where = "(autograd engine / no python frame)"
    for fr in ev.stack:
        if "fusiondepth_amd" in fr or "bench.py" in fr:
            where = fr.replace(ROOT + "/", ""); break
    sites[(ev.name, where)] += 1
print("| aten op (top level, views and allocations left out) | calls per step | first package frame |\n|---|---:|---|")
for (name, where), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    print("| %s | %d | %s |" % (name, n, where))

print("\n| op issuing a device copy / fill | input shapes | calls per step | parent chain |\n|---|---|---:|---|")
cp = collections.Counter()
for ev in prof.events():
    if ev.name not in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::add"):
        continue
    chain, p = [], ev.cpu_parent
    while p is not None and len(chain) < 4:
        chain.append(p.name); p = p.cpu_parent
    cp[(ev.name, str(ev.input_shapes)[:70], " < ".join(chain)[:110])] += 1
for (name, shp, chain), n in sorted(cp.items(), key=lambda kv: -kv[1])[:60]:
    print("| %s | %s | %d | %s |" % (name, shp, n, chain))
