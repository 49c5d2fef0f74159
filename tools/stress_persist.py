"""Stress: the persistent gather-GEMM against the per-tile kernel, bit for bit, many launches per shape with fresh
random data (a counted-wait or LDS hand-off race would show as a rare mismatch).  usage: stress_persist.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import hip_ops as ops, _lib
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
lib = _lib.load()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
SHAPES = [  # kind, N, Ci, H, W, Co, k, stride, pad_mode, bias, act
    ("convT", 8, 128, 128, 128, 64, 3, 2, "zero", True, None),
    ("conv", 8, 3, 256, 256, 64, 7, 1, "reflect", True, "lrelu"),
    ("conv", 12, 64, 128, 128, 64, 3, 1, "reflect", False, "relu"),
    ("conv", 3, 40, 150, 151, 192, 3, 1, "reflect", True, "lrelu"),
    ("conv", 10, 64, 120, 121, 128, 1, 1, "zero", True, None),
    ("convT", 6, 64, 80, 80, 128, 3, 2, "zero", False, None),
    ("conv", 16, 5, 256, 256, 64, 4, 2, "reflect", True, None),
]
bad = 0
# data gradient of a reflect-padded stride-1 layer: interior (persistent, zero padding) + ring GEMM + fold
for (N, Ci, H, Co) in ((8, 64, 128, 128), (12, 64, 128, 128)):
    w = (torch.randn(Co, Ci, 3, 3, device=dev) * (Ci * 9) ** -0.5).requires_grad_(False)
    n0 = lib.mt_kernel_variant_launches(0)
    mism = 0
    for it in range(max(iters // 3, 1)):
        x0 = ops.canon(torch.randn(N, Ci, H, H, device=dev))
        gy = None
        outs = []
        for on in (1, 0):
            lib.mt_kernel_variant_enable(0, on)
            x = x0.detach().requires_grad_()
            y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect", act="relu")
            if gy is None:
                gy = ops.canon(torch.randn_like(y.float())).detach()
            y.backward(gy)
            outs.append(x.grad.detach())
        if not torch.equal(outs[0], outs[1]):
            mism += 1
            d = (outs[0].float() - outs[1].float()).abs()
            print(f"  MISMATCH dgrad it {it}: {int((d > 0).sum())} elements differ, max {d.max().item():.3e}", flush=True)
    lib.mt_kernel_variant_enable(0, 1)
    print(f"dgrad N{N} {Ci}->{Co} {H}x{H}: {max(iters // 3, 1)} iterations, {lib.mt_kernel_variant_launches(0) - n0} persistent launches, "
          f"{mism} mismatches", flush=True)
    bad += mism
for kind, N, Ci, H, W, Co, k, st, pm, bias, act in SHAPES:
    w = torch.randn(*((Ci, Co, k, k) if kind == "convT" else (Co, Ci, k, k)), device=dev) * (Ci * k * k) ** -0.5
    b = torch.randn(Co, device=dev) * 0.1 if bias else None
    n0 = lib.mt_kernel_variant_launches(0)
    mism = 0
    for it in range(iters):
        x = ops.canon(torch.randn(N, Ci, H, W, device=dev))
        ys = []
        with torch.no_grad():
            for on in (1, 0):
                lib.mt_kernel_variant_enable(0, on)
                if kind == "conv":
                    ys.append(ops.conv2d(x, w, b, stride=st, pad=k // 2 if k > 1 else 0, pad_mode=pm, act=act))
                else:
                    ys.append(ops.conv_transpose2d(x, w, b, stride=st, pad=1, out_pad=1, act=act))
        if not torch.equal(ys[0], ys[1]):
            mism += 1
            d = (ys[0].float() - ys[1].float()).abs()
            print(f"  MISMATCH {kind} it {it}: {int((d > 0).sum())} elements differ, max {d.max().item():.3e}", flush=True)
    lib.mt_kernel_variant_enable(0, 1)
    used = lib.mt_kernel_variant_launches(0) - n0
    print(f"{kind} N{N} {Ci}->{Co} {H}x{W} k{k} s{st}: {iters} iterations, {used} persistent launches, {mism} mismatches", flush=True)
    bad += mism
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
