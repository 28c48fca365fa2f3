"""The fused stem tail alone (BatchNorm + ReLU + max-pool, forward and backward), n times: for the PMC passes.
usage: stem_tail_one.py batch [n] [want_feat 0|1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD
B = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 8; want = bool(int(sys.argv[3])) if len(sys.argv) > 3 else False
x = torch.randn(B, 64, 96, 320, device="cuda").requires_grad_(True)
bn = torch.nn.BatchNorm2d(64).cuda().train()
for _ in range(n):
    with FD.bn_groups(2):
        f0, p = FD.bn_relu_maxpool(x, bn, want_feature=want)
    loss = p.sum() + (f0.sum() if want else 0)
    torch.autograd.grad(loss, [x, bn.weight, bn.bias])
torch.cuda.synchronize()
