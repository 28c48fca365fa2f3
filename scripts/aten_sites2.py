"""ATen device work issued from Python during one optimiser step, by call site: a TorchDispatchMode records every aten op that is
not a view / allocation together with the innermost fusiondepth_amd frame.   (run on the GPU box)"""
import os, sys, collections, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer

opt = MonodepthOptions().parse(["--num_layers", "18", "--batch_size", "12", "--height", "192", "--width", "640", "--weights_init", "scratch"])
tr = Trainer(opt, verbose=False)
mbs = [synthetic.make_batch(tr.batch_size, 192, 640, seed=1234 + i) for i in range(tr.accumulate_step)]
inp = tr.stack_micro_batches(mbs)
for _ in range(3):
    tr.train_step(inp)
torch.cuda.synchronize()
SKIP = {"empty", "empty_like", "view", "slice", "select", "as_strided", "detach", "alias", "unsqueeze", "squeeze", "reshape", "expand",
        "permute", "t", "transpose", "_unsafe_view", "empty_strided", "unbind", "split", "narrow", "_local_scalar_dense", "is_same_size",
        "lift_fresh", "sym_size", "sym_numel", "sym_stride", "sym_storage_offset", "stride", "size", "numel", "record_stream"}
sites = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in SKIP:
            where = "?"
            for fr in reversed(traceback.extract_stack(limit=25)):
                if "fusiondepth_amd" in fr.filename and "aten_sites" not in fr.filename:
                    where = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name); break
            shp = ""
            for a in args:
                if torch.is_tensor(a):
                    shp = str(tuple(a.shape)); break
            sites[(name, where, shp)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    tr.train_step(inp)
torch.cuda.synchronize()
print("| aten op | calls | site | first tensor |\n|---|---:|---|---|")
tot = 0
for (name, where, shp), n in sorted(sites.items(), key=lambda kv: -kv[1]):
    tot += n
    print("| %s | %d | %s | %s |" % (name, n, where, shp))
print("total", tot)
