"""Patch-resident gather-GEMM (conv_patch_kernel.hip) on/off on the mid-size layers of the step (HIP events, random
data): forward and data gradient, us per call, and which of the two used the new kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import _lib
if os.environ.get("MT_DIAG_LIB"):          # diagnostic builds of the library (never the product path)
    _lib.LIB_PATH = os.path.abspath(os.environ["MT_DIAG_LIB"])
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
lib = _lib.load()

def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

LAYERS = [  # name, kind, N, Ci, H, Co, k, stride
    ("conv 3x3 s1 256->256 @32 N16", "conv", 16, 256, 32, 256, 3, 1),
    ("conv 3x3 s2 256->512 @32 N32", "conv", 32, 256, 32, 512, 3, 2),
    ("conv 4x4 s2 64->128 @128 N32", "conv4", 32, 64, 128, 128, 4, 2),
    ("conv 4x4 s2 128->256 @64 N32", "conv4", 32, 128, 64, 256, 4, 2),
    ("conv 3x3 s1 64->128 @128 N16", "conv", 16, 64, 128, 128, 3, 1),
    ("conv 3x3 s2 64->128 @256 N16", "conv", 16, 64, 256, 128, 3, 2),
    ("conv 3x3 s2 128->256 @128 N16", "conv", 16, 128, 128, 256, 3, 2),
    ("conv 3x3 s1 128->256 @64 N16", "conv", 16, 128, 64, 256, 3, 1),
    ("conv 3x3 s1 64->64 @128 N16", "conv", 16, 64, 128, 64, 3, 1),
    ("conv 3x3 s1 128->128 @64 N16", "conv", 16, 128, 64, 128, 3, 1),
    ("conv 7x7 s1 3->64 @256 N16", "conv", 16, 3, 256, 64, 7, 1),
    ("conv 4x4 s2 5->64 @256 N16", "conv", 16, 5, 256, 64, 4, 2),
    ("conv 3x3 s2 3->64 @256 N32", "conv", 32, 3, 256, 64, 3, 2),
    ("conv 3x3 s2 64->128 @128 N32", "conv", 32, 64, 128, 128, 3, 2),
    ("convT 3x3 s2 128->64 @128 N32", "convT", 32, 128, 128, 64, 3, 2),
    ("convT 3x3 s2 128->64 @128 N16", "convT", 16, 128, 128, 64, 3, 2),
    ("convT 3x3 s2 256->128 @64 N32", "convT", 32, 256, 64, 128, 3, 2),
    ("convT 3x3 s2 256->128 @64 N16", "convT", 16, 256, 64, 128, 3, 2),
]
only = sys.argv[1] if len(sys.argv) > 1 else None
print(f"{'layer':34s} {'fwd off':>8s} {'fwd on':>8s} {'dgrad off':>10s} {'dgrad on':>9s}")
tot = [0, 0, 0, 0]
for name, kind, N, Ci, H, Co, k, st in LAYERS:
    if only and only not in name: continue
    x = ops.canon(torch.randn(N, Ci, H, H, device=dev)).detach().requires_grad_()
    if kind == "conv":
        w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
        f = lambda: ops.conv2d(x, w, None, stride=st, pad=k // 2, pad_mode="reflect")
    elif kind == "conv4":
        w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
        f = lambda: ops.conv2d(x, w, None, stride=st, pad=1, pad_mode="zero", act="lrelu")
    else:
        w = torch.randn(Ci, Co, k, k, device=dev) * 0.05
        f = lambda: ops.conv_transpose2d(x, w, None, stride=st, pad=1, out_pad=1)
    y = f()
    gy = ops.canon(torch.randn_like(y.float())).detach()
    def fwd():
        with torch.no_grad(): f()
    def both():
        x.grad = None
        f().backward(gy)
    res = []
    for on in (0, 2):
        lib.mt_kernel_variant_enable(2, on)
        n0 = lib.mt_kernel_variant_launches(2)
        fwd()
        n1 = lib.mt_kernel_variant_launches(2)
        both()
        n2 = lib.mt_kernel_variant_launches(2)
        used = ("F" if n1 > n0 else "-") + ("B" if n2 - n1 > n1 - n0 else "-")
        tf = t(fwd); tb = t(both) - tf
        res += [tf, tb]
    lib.mt_kernel_variant_enable(2, 1)
    print(f"{name:34s} {res[0]:8.1f} {res[2]:8.1f} {res[1]:10.1f} {res[3]:9.1f}   patch kernel: {used}", flush=True)
    for i, v in enumerate((res[0], res[2], res[1], res[3])): tot[i] += v
print(f"{'sum':34s} {tot[0]:8.1f} {tot[1]:8.1f} {tot[2]:10.1f} {tot[3]:9.1f}")
