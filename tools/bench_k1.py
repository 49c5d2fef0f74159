import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = ops.canon(torch.randn(N, 256, 64, 64, device=dev))
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
with torch.no_grad():
    for _ in range(20):
        ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")
torch.cuda.synchronize()
