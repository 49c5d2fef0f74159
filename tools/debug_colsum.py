import sys; sys.path.insert(0,'.')
import torch, torch.nn.functional as F
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
for dt in (torch.float32, torch.bfloat16):
    ops.set_compute_dtype(dt)
    for (N,Ci,H,Co) in [(2,64,2,64),(2,32,4,64),(4,128,2,128),(2,8,8,16)]:
        torch.manual_seed(0)
        x = torch.randn(N,Ci,H,H); w = torch.randn(Co,Ci,3,3)*0.05; b = torch.randn(Co)*0.1
        xr,wr,br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
        yr = F.leaky_relu(F.conv2d(F.pad(xr,(1,1,1,1),mode='reflect'), wr, br, stride=2), 0.01)
        gy = torch.randn_like(yr); yr.backward(gy)
        for trial in range(3):
            xd,wd,bd = x.to(dev).requires_grad_(), w.to(dev).requires_grad_(), b.to(dev).requires_grad_()
            y = ops.conv2d(xd,wd,bd,stride=2,pad=1,pad_mode='reflect',act='lrelu')
            y.backward(gy.to(dev))
            e = (bd.grad.cpu()-br.grad).norm()/br.grad.norm()
            ew = (wd.grad.cpu()-wr.grad).norm()/wr.grad.norm()
            print(dt, (N,Ci,H,Co), 'db rel', f'{e.item():.2e}', 'dw rel', f'{ew.item():.2e}')
