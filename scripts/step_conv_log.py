"""One steady-state optimiser step of a BASELINE configuration with fd_tuning.log = 1: every convolution call with the kernel family it was
routed to (stderr; pipe through scripts/conv_log_summary.py).    python scripts/step_conv_log.py r18|r50|r18big|completor 2>&1 | python scripts/conv_log_summary.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic, tuning
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer

CFG = {"r50": (50, 192, 640, 8), "r18big": (18, 320, 1024, 8), "r18": (18, 192, 640, 12)}
name = sys.argv[1] if len(sys.argv) > 1 else "r18"
if name == "completor":
    from fusiondepth_amd.completor import Completor
    opt = MonodepthOptions().parse(["--weights_init", "scratch", "--batch_size", "12", "--completion_num_layers", "18"])
    tr = Completor(opt, verbose=False)
    H, W = opt.height, opt.width
else:
    layers, H, W, bs = CFG[name]
    opt = MonodepthOptions().parse(["--num_layers", str(layers), "--weights_init", "scratch", "--batch_size", str(bs), "--height", str(H), "--width", str(W)])
    tr = Trainer(opt, verbose=False)
mbs = [synthetic.make_batch(tr.batch_size, H, W, seed=1234 + i) for i in range(tr.accumulate_step)]
inp = tr.stack_micro_batches(mbs) if tr.stack_microbatches else mbs
for _ in range(4):
    tr.train_step(inp)
torch.cuda.synchronize()
tuning.set_lib(log=1)
tr.train_step(inp)
torch.cuda.synchronize()
tuning.set_lib(log=0)
