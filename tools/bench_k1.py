"""The K1 convolution (3x3 s1 256->256 on 64x64, reflection padding) through the C ABI, 20 launches of ONE of its three
GEMMs -- what tools/pmc_k1.sh profiles:   python tools/bench_k1.py [fwd|dgrad|wgrad] [N]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
mode = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "fwd"
N = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 16
x = ops.canon(torch.randn(N, 256, 64, 64, device=dev))
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
if mode == "fwd":
    with torch.no_grad():
        for _ in range(20):
            ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")
else:
    x.requires_grad_(mode == "dgrad")
    w.requires_grad_(mode == "wgrad")
    gy = None
    for _ in range(20):
        y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")
        if gy is None:
            gy = ops.canon(torch.randn_like(y.float())).detach()
        y.backward(gy)
        x.grad = None
        w.grad = None
torch.cuda.synchronize()
