import sys
sys.path.insert(0, '.')
import torch
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
x = ops.canon(torch.randn(16, 256, 64, 64, device=dev))
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
def t(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
with torch.no_grad():
    print("plain  %.1f us" % t(lambda: ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")))
    print("stats  %.1f us" % t(lambda: ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect", stats=True)))
    y, s = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect", stats=True)
    print("norm with sums %.1f us" % t(lambda: ops.instance_norm_act(y, act="relu", sums=s)))
    print("norm w/o  sums %.1f us" % t(lambda: ops.instance_norm_act(y, act="relu")))
