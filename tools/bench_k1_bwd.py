import sys
sys.path.insert(0, '.')
import torch
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
x = ops.canon(torch.randn(16, 256, 64, 64, device=dev)).detach().requires_grad_()
w = (torch.randn(256, 256, 3, 3, device=dev) * 0.05).requires_grad_()
y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")
gy = ops.canon(torch.randn_like(y.float())).detach()
for _ in range(10):
    x.grad = None; w.grad = None
    y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect"); y.backward(gy)
torch.cuda.synchronize()
