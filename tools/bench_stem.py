"""7x7 stem forward: direct kernel on / off (HIP events, random data), with and without the fused statistics."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import hip_ops as ops, _lib
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
lib = _lib.load()
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for N in (16, 32):
    x = ops.canon(torch.randn(N, 3, 256, 256, device=dev))
    w = torch.randn(64, 3, 7, 7, device=dev) * 0.08
    b = torch.randn(64, device=dev) * 0.1
    for st in (False, True):
        def f():
            with torch.no_grad(): ops.conv2d(x, w, b, stride=1, pad=3, pad_mode="reflect", stats=st)
        r = []
        for on in (0, 1):
            lib.mt_kernel_variant_enable(1, on); r.append(t(f))
        lib.mt_kernel_variant_enable(1, 1)
        print(f"N{N} stats={int(st)}: gather-GEMM {r[0]:.1f} us, direct {r[1]:.1f} us")

for N in (16, 32):
    x = ops.canon(torch.randn(N, 3, 256, 256, device=dev))
    w = (torch.randn(64, 3, 7, 7, device=dev) * 0.08).requires_grad_()
    gy = None
    r = []
    for on in (0, 1):
        lib.mt_kernel_variant_enable(1, on)
        def fwd():
            with torch.no_grad(): ops.conv2d(x, w, None, stride=1, pad=3, pad_mode="reflect")
        def both():
            w.grad = None
            y = ops.conv2d(x, w, None, stride=1, pad=3, pad_mode="reflect")
            y.backward(gy if gy is not None else torch.ones_like(y))
        y0 = ops.conv2d(x, w, None, stride=1, pad=3, pad_mode="reflect")
        gy = ops.canon(torch.randn_like(y0.float())).detach()
        r.append(t(both) - t(fwd))
    lib.mt_kernel_variant_enable(1, 1)
    print(f"N{N} weight gradient (backward minus forward, x without gradient): gather form {r[0]:.1f} us, direct {r[1]:.1f} us")

for N in (16,):
    x = ops.canon(torch.randn(N, 3, 256, 256, device=dev)).detach().requires_grad_()
    w = torch.randn(64, 3, 7, 7, device=dev) * 0.08
    r = []
    y0 = ops.conv2d(x, w, None, stride=1, pad=3, pad_mode="reflect")
    gy = ops.canon(torch.randn_like(y0.float())).detach()
    for on in (0, 1):
        lib.mt_kernel_variant_enable(1, on)
        def fwd():
            with torch.no_grad(): ops.conv2d(x, w, None, stride=1, pad=3, pad_mode="reflect")
        def both():
            x.grad = None
            ops.conv2d(x, w, None, stride=1, pad=3, pad_mode="reflect").backward(gy)
        r.append(t(both) - t(fwd))
    lib.mt_kernel_variant_enable(1, 1)
    print(f"N{N} data gradient (backward minus forward, weight without gradient): gather form {r[0]:.1f} us, direct {r[1]:.1f} us")
