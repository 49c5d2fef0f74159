"""A/B of the accumulator-stationary weight gradient (wgrad_rows_kernel.hip) against the 128 x 128 tile kernel it replaces, interleaved
rounds in one process: weight gradient (GEMM + slab sum) of the 3x3 layers between the stem and the bottleneck at the bench's shapes.
The time of a call is (forward + weight gradient) - forward, HIP events; the input needs no gradient, so no data gradient runs.

    python tools/bench_wgrad_rows.py [rounds]
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
    ("Ec down 3x3s2 128->256 @128 N16", "conv", 16, 128, 128, 128, 256, 2, "reflect"),
    ("Es 3x3s1 64->64 @128 N16", "conv", 16, 64, 128, 128, 64, 1, "reflect"),
    ("Es 3x3s1 64->128 @128 N16", "conv", 16, 64, 128, 128, 128, 1, "reflect"),
    ("Es 3x3s1 128->128 @64 N16", "conv", 16, 128, 64, 64, 128, 1, "reflect"),
    ("Es 3x3s1 128->256 @64 N16", "conv", 16, 128, 64, 64, 256, 1, "reflect"),
    ("Dec convT 256->128 @64 N16", "convT", 16, 256, 64, 64, 128, 2, "zero"),
    ("Dec convT 256->128 @64 N32", "convT", 32, 256, 64, 64, 128, 2, "zero"),
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
    only = int(os.environ.get("MT_BENCH_ONLY", "-1"))          # (one arm only: for rocprofv3)
    for name, kind, N, Ci, H, W, Co, stride, mode in SHAPES:
        x = ops.canon(torch.randn(N, Ci, H, W, device=dev)).detach()
        wshape = (Co, Ci, 3, 3) if kind == "conv" else (Ci, Co, 3, 3)
        w = (torch.randn(*wshape, device=dev) * 0.05).requires_grad_()

        def fwd():
            if kind == "conv":
                return ops.conv2d(x, w, None, stride=stride, pad=1, pad_mode=mode)
            return ops.conv_transpose2d(x, w, None, stride=stride, pad=1, out_pad=1)
        y = fwd()
        gy = ops.canon(torch.randn_like(y.float())).detach()

        def f_only():
            with torch.no_grad():
                fwd()

        def fb():
            w.grad = None
            fwd().backward(gy)
        res = {0: [], 1: []}
        arms = (0, 1) if only < 0 else (only,)
        for _ in range(rounds):
            for on in arms:
                lib.mt_kernel_variant_enable(5, on)
                n0 = lib.mt_kernel_variant_launches(5)
                tf = timeit(f_only)
                tb = timeit(fb) - tf
                res[on].append((tb, lib.mt_kernel_variant_launches(5) - n0))
        lib.mt_kernel_variant_enable(5, 1)
        flops = 2.0 * y.numel() // y.shape[1] * Ci * Co * 9 if kind == "conv" else 2.0 * x.numel() // x.shape[1] * Ci * Co * 9
        for on in arms:
            tb = sorted(r[0] for r in res[on])[len(res[on]) // 2]
            print(f"{name:34s} rows={on} wgrad + slab sum {tb:7.1f} us ({flops / tb / 1e6:6.0f} TFLOP/s) | launches on the row walker {res[on][0][1]}",
                  flush=True)


if __name__ == "__main__":
    main()
