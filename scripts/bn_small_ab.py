import os, sys, subprocess, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fusiondepth_amd import functional as FD
torch.manual_seed(0)
def run():
    out = {}
    for (N, C, H, W, G, relu, res) in [(2, 64, 16, 24, 1, True, False), (4, 128, 8, 12, 2, True, True), (2, 64, 32, 48, 1, False, False), (2, 256, 4, 6, 1, True, True), (2, 256, 4, 6, 1, False, False), (2, 64, 32, 48, 1, True, False), (2, 128, 8, 12, 1, True, True), (12, 512, 6, 20, 2, True, True), (24, 256, 12, 40, 4, True, False)]:
        bn = torch.nn.BatchNorm2d(C).cuda()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        x = torch.randn(N, C, H, W, device="cuda", requires_grad=True)
        r = torch.randn(N, C, H, W, device="cuda", requires_grad=True) if res else None
        with FD.bn_groups(G):
            y = FD.batch_norm(x, bn, residual=r, relu=relu)
        cot = torch.randn_like(y)
        gs = torch.autograd.grad((y * cot).sum(), [x, bn.weight, bn.bias] + ([r] if res else []))
        out[(N, C, H, W, G)] = [y.detach().double().cpu()] + [g.double().cpu() for g in gs] + [bn.running_mean.double().cpu(), bn.running_var.double().cpu()]
    return out
if len(sys.argv) > 1:
    torch.save(run(), sys.argv[1]); sys.exit()
a = run()
env = dict(os.environ, FD_BN_SMALL_OFF="1")
subprocess.check_call([sys.executable, __file__, "/tmp/bn_big.pt"], env=env)
b = torch.load("/tmp/bn_big.pt")
for k in a:
    print(k, ["%.2e" % float((u - v).abs().max() / (v.abs().max() + 1e-30)) for u, v in zip(a[k], b[k])])
