"""A/B of the weight-stationary gather-GEMM (conv_wsreg_kernel.hip) against the tile kernels it replaces, interleaved rounds in one
process on random data (cdna guide rule 24): forward and data gradient of the short-K layers of the step at the bench's shapes.

    python tools/bench_wsreg.py [rounds]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from masterthesis_amd import _lib as L, hip_ops as ops

dev = torch.device("cuda:0")
ops.set_compute_dtype(torch.bfloat16)
lib = L.load()

# name, kind, N, Ci, H, W, Co, stride, pad_mode
SHAPES = [
    ("Ec down 3x3s2 64->128 @256 N16", "conv", 16, 64, 256, 256, 128, 2, "reflect"),
    ("Es 3x3s1 64->64 @128 N16", "conv", 16, 64, 128, 128, 64, 1, "reflect"),
    ("Es 3x3s1 64->128 @128 N16", "conv", 16, 64, 128, 128, 128, 1, "reflect"),
    ("Es 3x3s1 128->128 @64 N16", "conv", 16, 128, 64, 64, 128, 1, "reflect"),
    ("Dec convT 128->64 @128 N16", "convT", 16, 128, 128, 128, 64, 2, "zero"),
    ("Dec convT 128->64 @128 N32", "convT", 32, 128, 128, 128, 64, 2, "zero"),
]


def timeit(fn, iters=20):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    for name, kind, N, Ci, H, W, Co, stride, mode in SHAPES:
        x = ops.canon(torch.randn(N, Ci, H, W, device=dev)).detach().requires_grad_()
        wshape = (Co, Ci, 3, 3) if kind == "conv" else (Ci, Co, 3, 3)
        w = (torch.randn(*wshape, device=dev) * 0.05).requires_grad_()
        b = torch.randn(Co, device=dev) * 0.1

        def fwd():
            if kind == "conv":
                return ops.conv2d(x, w, b, stride=stride, pad=1, pad_mode=mode)
            return ops.conv_transpose2d(x, w, b, stride=stride, pad=1, out_pad=1)
        y = fwd()
        gy = ops.canon(torch.randn_like(y.float())).detach()

        def f_only():
            with torch.no_grad():
                fwd()

        def fb():
            x.grad = None
            w.grad = None
            fwd().backward(gy)
        res = {0: [], 1: []}
        for _ in range(rounds):
            for on in (0, 1):
                lib.mt_kernel_variant_enable(4, on)
                n0 = lib.mt_kernel_variant_launches(4)
                tf = timeit(f_only)
                tb = timeit(fb) - tf
                res[on].append((tf, tb, lib.mt_kernel_variant_launches(4) - n0))
        lib.mt_kernel_variant_enable(4, 1)
        out_b = y.numel() // y.shape[1] * ops.padc(y.shape[1]) * 2
        in_b = x.numel() // x.shape[1] * ops.padc(x.shape[1]) * 2
        for on in (0, 1):
            tf = sorted(r[0] for r in res[on])[len(res[on]) // 2]
            tb = sorted(r[1] for r in res[on])[len(res[on]) // 2]
            print(f"{name:34s} wsreg={on} fwd {tf:7.1f} us ({(in_b + out_b) / tf / 1e6:5.2f} TB/s algorithmic) | dgrad+wgrad {tb:7.1f} us"
                  f" | wsreg launches per timed call set {res[on][0][2]}", flush=True)


if __name__ == "__main__":
    main()
