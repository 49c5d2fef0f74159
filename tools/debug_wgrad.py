import torch, sys
sys.path.insert(0, '.')
from masterthesis_amd import hip_ops as ops
ops.set_compute_dtype(torch.bfloat16)
dev = torch.device('cuda:0')
N,H,W,Ci,Co = 1,8,8,8,8
x = torch.zeros(N,Ci,H,W)
x[0,0] = torch.arange(1,65).float().view(8,8)
x[0,1] = 100+torch.arange(1,65).float().view(8,8)
w = torch.zeros(Co,Ci,1,1)
res=[]
for m0 in range(64):
    xd = x.to(dev).requires_grad_(); wd = w.to(dev).requires_grad_()
    y = ops.conv2d(xd, wd)
    gy = torch.zeros(N,Co,H,W); gy.view(Co,64)[1,m0] = 1.0
    y.backward(gy.to(dev))
    g = wd.grad.cpu().view(Co,Ci)
    res.append((m0+1, g[1,0].item(), g[1,1].item(), g.abs().sum().item()))
for r in res: print(r)
