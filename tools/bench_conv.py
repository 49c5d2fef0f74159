"""Micro-benchmark of the conv kernels on the dominant shapes (HIP events, random data)."""
import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
def bench(name, N, Ci, H, W, Co, k, stride, pad, mode, iters=30):
    x = ops.canon(torch.randn(N, Ci, H, W, device=dev)).detach().requires_grad_()
    w = (torch.randn(Co, Ci, k, k, device=dev) * 0.05).requires_grad_()
    y = ops.conv2d(x, w, None, stride=stride, pad=pad, pad_mode=mode)
    gy = ops.canon(torch.randn_like(y.float())).detach()
    def fwd():
        with torch.no_grad(): ops.conv2d(x, w, None, stride=stride, pad=pad, pad_mode=mode)
    flop = 2.0 * y.shape[0] * y.shape[2] * y.shape[3] * Co * Ci * k * k
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
    tf = t(fwd)
    def bwd_both():
        x.grad = None; w.grad = None
        yy = ops.conv2d(x, w, None, stride=stride, pad=pad, pad_mode=mode); yy.backward(gy)
    tb = t(bwd_both) - tf
    print(f"{name:28s} fwd {tf*1e3:7.1f} us {flop/tf/1e9:7.0f} TF | dgrad+wgrad {tb*1e3:7.1f} us {2*flop/tb/1e9:7.0f} TF")
bench("K1 3x3 256->256 @64 N16", 16, 256, 64, 64, 256, 3, 1, 1, "reflect")
bench("K1 half N8", 8, 256, 64, 64, 256, 3, 1, 1, "reflect")
bench("down 3x3s2 128->256 @128", 16, 128, 128, 128, 256, 3, 2, 1, "reflect")
bench("down 3x3s2 64->128 @256", 16, 64, 256, 256, 128, 3, 2, 1, "reflect")
bench("stem 7x7 3->64 @256", 16, 3, 256, 256, 64, 7, 1, 3, "reflect")
bench("D 3x3s2 512->1024 @16", 32, 512, 16, 16, 1024, 3, 2, 1, "reflect")
