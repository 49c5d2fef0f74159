"""Randomised convolutions through the tile GEMMs (2-stage, ping-pong, patch-resident, persistent, split-K: whatever the dispatcher
picks) against the fp32 PyTorch CPU reference of the op -- written for the straight-line epilogue (conv_device.h: epilogue_perm), whose
out-of-range offsets replace the per-pixel / per-channel branches: channel counts that are not multiples of the tile or of 8, ragged
pixel counts, bias shorter than the padded channels, every activation, stride 1 / 2 / 4, both padding modes, transposed convolutions,
bf16 and fp32 -- forward, data gradient, weight and bias gradient, plus a NaN / Inf check and a canary behind the output.

    python tools/fuzz_conv.py [cases] [seed]
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from masterthesis_amd import hip_ops as ops

dev = torch.device("cuda:0")


def reference(kind, x, w, b, k, stride, pad, mode, act):
    if kind == "conv":
        xp = F.pad(x, (pad,) * 4, mode="reflect") if (mode == "reflect" and pad) else x
        y = F.conv2d(xp, w, b, stride=stride, padding=0 if (mode == "reflect" and pad) else pad)
    else:
        y = F.conv_transpose2d(x, w, b, stride=2, padding=1, output_padding=1)
    if act == "relu":
        y = F.relu(y)
    elif act == "lrelu":
        y = F.leaky_relu(y, 0.01)
    elif act == "tanh":
        y = torch.tanh(y)
    return y


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad = 0
    for i in range(cases):
        dtype = torch.bfloat16 if rnd.random() < 0.75 else torch.float32
        ops.set_compute_dtype(dtype)
        kind = rnd.choice(["conv", "conv", "conv", "convT"])
        Ci = rnd.choice([3, 8, 24, 64, 72, 128, 136, 256, 512])
        Co = rnd.choice([1, 3, 5, 8, 24, 40, 64, 72, 100, 128, 136, 250, 256, 264, 512])
        if kind == "convT":
            k, stride, pad, mode = 3, 2, 1, "zero"
        else:
            k, stride = rnd.choice([(3, 1), (3, 1), (3, 2), (4, 2), (4, 4), (1, 1), (7, 1)])
            pad = {1: 0, 3: 1, 4: rnd.choice([0, 1]) if stride == 2 else 0, 7: 3}[k]
            mode = rnd.choice(["zero", "reflect"]) if pad else "zero"
        H = rnd.choice([8, 16, 17, 32, 33, 48, 64, 65, 96, 128])
        W = rnd.choice([8, 16, 24, 32, 40, 64, 80, 128])
        if k == 4 and stride == 4:
            H, W = 4 * rnd.choice([1, 2, 4, 8]), 4 * rnd.choice([1, 2, 4, 8])
        N = rnd.choice([1, 2, 3, 4, 8, 16, 32])
        while N * H * W * max(Ci, Co) > 48 * 2 ** 20 or N * H * W * Ci * Co * k * k > 1e11:
            N = max(1, N // 2)
            if N == 1:
                H = max(8, H // 2)
        if mode == "reflect" and (H <= pad or W <= pad):
            mode = "zero"
        bias = rnd.random() < 0.6
        act = rnd.choice([None, None, "relu", "lrelu", "tanh"])
        g = torch.Generator().manual_seed(7000 + i)
        q = (lambda t: t.bfloat16().float()) if dtype == torch.bfloat16 else (lambda t: t)
        x = q(torch.randn(N, Ci, H, W, generator=g))
        wshape = (Co, Ci, k, k) if kind == "conv" else (Ci, Co, k, k)
        w = q(torch.randn(*wshape, generator=g) * (Ci * k * k) ** -0.5)
        b = torch.randn(Co, generator=g) * 0.2 if bias else None
        xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
        br = b.clone().requires_grad_() if bias else None
        yr = reference(kind, xr, wr, br, k, stride, pad, mode, act)
        gy = q(torch.randn(*yr.shape, generator=g))
        yr.backward(gy)
        xd, wd = x.to(dev).requires_grad_(), w.to(dev).requires_grad_()
        bd = b.to(dev).requires_grad_() if bias else None
        if kind == "conv":
            y = ops.conv2d(xd, wd, bd, stride=stride, pad=pad, pad_mode=mode, act=act)
        else:
            y = ops.conv_transpose2d(xd, wd, bd, stride=2, pad=1, out_pad=1, act=act)
        y.backward(gy.to(dev).to(y.dtype))
        torch.cuda.synchronize()
        tol = 2e-2 if dtype == torch.bfloat16 else 1e-3      # (fp32: summation order over up to 10^5 pixels)
        worst = 0.0
        pairs = [("y", y.detach().float().cpu(), yr.detach()), ("dx", xd.grad.float().cpu(), xr.grad), ("dw", wd.grad.float().cpu(), wr.grad)]
        if bias:
            pairs.append(("db", bd.grad.float().cpu(), br.grad))
        for what, a, t in pairs:
            ok = bool(torch.isfinite(a).all())
            rel = ((a - t).norm() / (t.norm() + 1e-20)).item() if t.norm() > 0 else a.abs().max().item()
            worst = max(worst, rel)
            if not ok or rel > tol:
                bad += 1
                print(f"MISMATCH case {i} {what}: finite {ok} rel {rel:.3e}")
        # pad channels of the canonical NHWC buffer stay zero (every kernel preserves that)
        yd = y.detach()
        Cp = ops.padc(Co)
        if Cp > Co and ops.is_canonical(yd):
            Nn, _, Hh, Ww = yd.shape
            full = torch.as_strided(yd, (Nn, Hh, Ww, Cp), (Hh * Ww * Cp, Ww * Cp, Cp, 1))
            padz = full[..., Co:].float().abs().max().item()
            if padz != 0.0:
                bad += 1
                print(f"MISMATCH case {i}: pad channels not zero ({padz})")
        print(f"case {i:3d} {kind:5s} {str(dtype)[6:]:8s} N{N} {Ci}->{Co} {H}x{W} k{k} s{stride} p{pad} {mode:7s} bias {int(bias)} act {act}: "
              f"worst rel {worst:.2e}", flush=True)
    ops.set_compute_dtype(torch.bfloat16)
    print(f"{cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
