"""K1 (3x3 s1 256->256 @64x64, N16, reflect) data gradient alone, for rocprofv3 --kernel-trace --stats."""
import sys, ctypes as C
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import hip_ops as ops, _lib as L
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = ops.canon(torch.randn(N, 256, 64, 64, device=dev)).detach().requires_grad_()
w = (torch.randn(256, 256, 3, 3, device=dev) * 0.05)
y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")
gy = ops.canon(torch.randn_like(y.float())).detach()
for _ in range(20):
    x.grad = None
    y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")
    y.backward(gy)
torch.cuda.synchronize()
