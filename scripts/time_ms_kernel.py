"""Time ONLY fd_photo_ms_fwd (main kernel + fin) with HIP events over many launches (no autograd glue)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, synthetic, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ROWS = int(sys.argv[2]) if len(sys.argv) > 2 else 0
GRAD = int(sys.argv[3]) if len(sys.argv) > 3 else 1
H, W = 192, 640
batch = synthetic.make_batch(B, H, W, seed=1)
po = FD.PhotoOptions()
tgt = batch[("color", 0, 0)]
srcs = [batch[("color", -1, 0)], batch[("color", 1, 0)]]
ident = torch.empty(B, 2, H, W, device="cuda")
for i, s_ in enumerate(srcs):
    FD.reprojection_loss_map(s_, tgt, True, out=ident[:, i:i + 1])
I = torch.eye(4, device="cuda").repeat(B, 1, 1); I[:, 0, 3] = 0.05
disps = [torch.rand(B, 1, H >> s, W >> s, device="cuda").mul_(0.1).add_(0.02).requires_grad_(bool(GRAD)) for s in range(4)]
noise = torch.randn(4, B, 2, H, W, device="cuda")
def fwd():
    return FD.photo_loss_ms(disps, [I, I], batch[("K", 0)], batch[("inv_K", 0)], srcs, tgt, ident, list(noise), batch["4beam"], (0, 1, 2, 3), po, 2, ROWS)
for _ in range(3): fwd()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 30
a.record()
for _ in range(N): fwd()
b.record(); torch.cuda.synchronize()
print("%s rows=%d grad=%d: %.1f us per fd_photo_ms_fwd (incl. 2 proj-matrix + fin launches, host-paced)" % (os.path.basename(_lib.LIB_PATH), ROWS, GRAD, a.elapsed_time(b) * 1e3 / N))
