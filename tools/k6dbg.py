import sys, os, torch
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "flash-molecular-dynamics_amd")]
import flashmd_amd.kernels as K
g = torch.Generator().manual_seed(6)
m, k = 257, 128
go, yy = torch.randn(m, k, generator=g).half(), (torch.rand(m, k, generator=g) * 2 - 1).half()
W = torch.eye(k).half()
out = K.fused_tanh_backward_grad_x_v2(go.cuda(), yy.cuda(), W.cuda()).cpu()
gz = (go.float() * (1 - yy.float() ** 2)).half().float()
bad = (out != gz).nonzero()
print("mismatches", bad.shape[0], "of", m * k)
for i, j in bad[:8].tolist():
    gv, yv = go[i, j].float().item(), yy[i, j].float().item()
    exact = gv * (1 - yv * yv)
    print(i, j, "g", gv, "y", yv, "hip", out[i, j].item(), "torch", gz[i, j].item(), "fp64 exact", exact)
