"""The bench.py loss-path probe on its own (for rocprofv3): 4 scales of fused photometric + SI loss, forward + backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
H, W = 192, 640
batch = synthetic.make_batch(B, H, W, seed=1)
po = FD.PhotoOptions()
tgt = batch[("color", 0, 0)]
srcs = [batch[("color", -1, 0)], batch[("color", 1, 0)]]
ident = torch.empty(B, 2, H, W, device="cuda")
for i, s_ in enumerate(srcs):
    FD.reprojection_loss_map(s_, tgt, True, out=ident[:, i:i + 1])
I = torch.eye(4, device="cuda").repeat(B, 1, 1); I[:, 0, 3] = 0.05
disps = [torch.rand(B, 1, H >> s, W >> s, device="cuda").mul_(0.1).add_(0.02).requires_grad_(True) for s in range(4)]
noise = torch.randn(B, 2, H, W, device="cuda")
def step():
    tot = 0
    for s in range(4):
        photo, si = FD.photo_loss(disps[s], [I, I], batch[("K", 0)], batch[("inv_K", 0)], srcs, tgt, ident, noise, batch["4beam"], po)[:2]
        tot = tot + photo + si
    tot.backward()
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 10): step()
torch.cuda.synchronize(); print("done")
