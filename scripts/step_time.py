"""Step time of the default training step (ResNet-18, 640x192, --batch_size 12) without bench.py's checks: median of three 20-step
windows.  For what-if builds (FD_LIBFDHIP=...libfdhip_skip.so FD_SKIP=<sites>, scripts/build_skip_ablation.sh) whose results are wrong
by design.  usage: step_time.py [label]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import synthetic
from fusiondepth_amd.options import MonodepthOptions
from fusiondepth_amd.trainer import Trainer
opt = MonodepthOptions().parse(["--num_layers", "18", "--weights_init", "scratch", "--batch_size", "12", "--height", "192", "--width", "640"])
_shift = [torch.cuda.Stream() for _ in range(int(os.environ.get("FD_STREAM_SHIFT", "0")))]     # moves the step's streams along the runtime's stream -> hardware-queue round robin
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
win, host = [], []
for w in range(3):
    t0 = time.perf_counter()
    for i in range(20):
        tr.train_step(pool[i % 6])
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    win.append((time.perf_counter() - t0) * 50); host.append(th * 50)
o = sorted(range(3), key=lambda i: win[i])[1]
print("%-28s step %.2f ms (windows %s), host issue %.2f ms" % (sys.argv[1] if len(sys.argv) > 1 else os.environ.get("FD_SKIP", "-"),
      win[o], " ".join("%.2f" % x for x in win), host[o]), flush=True)
