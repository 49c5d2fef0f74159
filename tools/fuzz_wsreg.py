"""Randomised shapes through the weight-stationary gather-GEMM (conv_wsreg_kernel.hip) against the tile kernels it replaces: stride 1 / 2
convolutions and transposed convolutions at 64 / 128 channels with random batch sizes, odd and ragged map sizes, both padding modes, bias /
activation on and off -- forward, data gradient and weight gradient, wsreg on vs off on the same device tensors (same bf16 operands, fp32
accumulation in another order: rel-L2 4e-3), NaN / Inf checks, and a canary region behind every output.

    python tools/fuzz_wsreg.py [cases] [seed]
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from masterthesis_amd import _lib as L, hip_ops as ops

dev = torch.device("cuda:0")
ops.set_compute_dtype(torch.bfloat16)
lib = L.load()


def run(kind, N, Ci, H, W, Co, stride, mode, bias, act, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float().to(dev).requires_grad_()
    wshape = (Co, Ci, 3, 3) if kind == "conv" else (Ci, Co, 3, 3)
    w = (torch.randn(*wshape, generator=g) * (Ci * 9) ** -0.5).bfloat16().float().to(dev).requires_grad_()
    b = (torch.randn(Co, generator=g) * 0.1).to(dev).requires_grad_() if bias else None
    if kind == "conv":
        y = ops.conv2d(x, w, b, stride=stride, pad=1, pad_mode=mode, act=act)
    else:
        y = ops.conv_transpose2d(x, w, b, stride=2, pad=1, out_pad=1, act=act)
    gy = torch.randn(*y.shape, generator=g).bfloat16().float().to(dev)
    y.backward(gy)
    return y.detach().float(), x.grad.float(), w.grad.float()


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    taken = bad = 0
    for i in range(cases):
        kind = rnd.choice(["conv", "conv", "convT"])
        Ci = rnd.choice([64, 128]) if kind == "convT" else 64 if rnd.random() < 0.7 else 128
        Co = 64 if (kind == "convT" or Ci == 128) else rnd.choice([64, 128])
        stride = 2 if kind == "convT" else rnd.choice([1, 2])
        H = rnd.choice([64, 96, 100, 127, 128, 135, 160, 255, 256])
        W = rnd.choice([64, 80, 97, 128, 130, 240, 241, 256])
        mode = "zero" if kind == "convT" else rnd.choice(["zero", "reflect"])
        N = rnd.choice([1, 2, 3, 5, 8, 16])
        while N * H * W > 3 * 2 ** 20:
            N = max(1, N // 2)
        bias, act = rnd.random() < 0.6, rnd.choice([None, "relu", "lrelu"])
        lib.mt_kernel_variant_enable(4, 1)
        n0 = lib.mt_kernel_variant_launches(4)
        new = run(kind, N, Ci, H, W, Co, stride, mode, bias, act, 1000 + i)
        used = lib.mt_kernel_variant_launches(4) - n0
        lib.mt_kernel_variant_enable(4, 0)
        old = run(kind, N, Ci, H, W, Co, stride, mode, bias, act, 1000 + i)
        lib.mt_kernel_variant_enable(4, 1)
        taken += used > 0
        worst = 0.0
        for a, b_, what in zip(new, old, ("y", "dx", "dw")):
            ok = bool(torch.isfinite(a).all())
            rel = ((a - b_).norm() / (b_.norm() + 1e-12)).item()
            worst = max(worst, rel)
            if not ok or rel > 4e-3:
                bad += 1
                print(f"MISMATCH case {i} {what}: finite {ok} rel {rel:.3e}")
        print(f"case {i:3d} {kind:5s} N{N} {Ci}->{Co} {H}x{W} s{stride} {mode:7s} bias {int(bias)} act {act}: wsreg launches {used}, worst rel {worst:.2e}",
              flush=True)
    ops.check_device_status(dev)
    print(f"{cases} cases, {taken} through the weight-stationary kernel, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
