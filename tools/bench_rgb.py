import sys
sys.path.insert(0, '.')
import torch
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
x = ops.canon(torch.randn(16, 64, 256, 256, device=dev)).detach().requires_grad_()
w = (torch.randn(64, 3, 1, 1, device=dev) * 0.05).requires_grad_()
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
def fwd():
    with torch.no_grad(): ops.conv_transpose2d(x, w, None, stride=1, pad=0, out_pad=0, act="tanh")
y = ops.conv_transpose2d(x, w, None, stride=1, pad=0, out_pad=0, act="tanh")
gy = ops.canon(torch.randn_like(y.float())).detach()
def both():
    x.grad = None; w.grad = None
    yy = ops.conv_transpose2d(x, w, None, stride=1, pad=0, out_pad=0, act="tanh"); yy.backward(gy)
tf = t(fwd); tb = t(both)
print(f"toRGB fwd {tf:.1f} us, fwd+bwd {tb:.1f} us (bwd {tb-tf:.1f})")
