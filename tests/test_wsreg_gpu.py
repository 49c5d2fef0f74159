"""The weight-stationary gather-GEMM of round 4 (csrc/conv_wsreg_kernel.hip: a layer's weights live in the waves' registers, the
tile's input patch in LDS, one barrier per tile) against the fp32 CPU reference of the op and against the tile kernels it
replaces.  Shapes are the short-K / huge-M layers between the stem and the 256-channel bottleneck (reference networks.py:33,248;
blocks.py:73,93-119): stride-1 and stride-2 gathers at 64 input channels, the transposed convolution 128 -> 64 (scatter form: four
sub-pixel phases from one patch) and their data gradients, with reflection / zero padding, ragged and odd-sized maps (the
540 x 960 sampling path), bias + activation epilogues and the fused InstanceNorm statistics.  Every case asserts that the launches
it is about really ran on the new kernel (mt_kernel_variant_launches(4))."""
import pytest
import torch
import torch.nn.functional as F

from test_persist_gpu import case_seed, check_against_reference

pytestmark = pytest.mark.gpu

# name, kind, N, Ci, H, W, Co, k, stride, pad, pad_mode, bias, act, (wsreg launches expected in fwd, in the data gradient)
CASES = [
    ("g_s1_64_64_reflect", "conv", 6, 64, 128, 128, 64, 3, 1, 1, "reflect", True, "lrelu", (1, None)),
    ("g_s1_64_128_reflect", "conv", 4, 64, 128, 128, 128, 3, 1, 1, "reflect", True, None, (1, None)),
    ("g_s2_64_128_reflect", "conv", 4, 64, 256, 256, 128, 3, 2, 1, "reflect", True, "relu", (1, 1)),       # dgrad: scatter 128 -> 64
    ("g_s1_64_64_zero_ragged", "conv", 8, 64, 100, 120, 64, 3, 1, 1, "zero", False, None, (1, None)),
    ("g_s2_64_128_zero_odd", "conv", 6, 64, 135, 241, 128, 3, 2, 1, "zero", True, None, (1, 1)),
    ("g_s1_128_128_not_taken", "conv", 16, 128, 64, 64, 128, 3, 1, 1, "reflect", True, "lrelu", (0, None)),  # (8 x 16 channels: no gain)
    ("g_s1_128_64_zero", "conv", 8, 128, 96, 96, 64, 3, 1, 1, "zero", False, None, (1, None)),
    ("s_convT_128_64", "convT", 4, 128, 128, 128, 64, 3, 2, 1, "zero", True, None, (1, 1)),                # dgrad: gather s2 64 -> 128
    ("s_convT_128_64_odd_relu", "convT", 2, 128, 135, 240, 64, 3, 2, 1, "zero", True, "relu", (1, 1)),      # dec2.1 at 540 x 960
    ("g_s2_64_128_zero", "conv", 4, 64, 256, 256, 128, 3, 2, 1, "zero", False, "lrelu", (1, 1)),          # dgrad: the e = 1 scatter pattern
]


def _run(ops, lib, case, dev):
    name, kind, N, Ci, H, W, Co, k, stride, pad, pad_mode, bias, act, _ = case
    g = torch.Generator().manual_seed(case_seed(name))
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    wshape = (Co, Ci, k, k) if kind == "conv" else (Ci, Co, k, k)
    w = (torch.randn(*wshape, generator=g) * (Ci * k * k) ** -0.5).bfloat16().float()
    b = (torch.randn(Co, generator=g) * 0.1) if bias else None
    xd = x.to(dev).requires_grad_()
    wd = w.to(dev).requires_grad_()
    bd = b.to(dev).requires_grad_() if bias else None
    n0 = lib.mt_kernel_variant_launches(4)
    if kind == "conv":
        y = ops.conv2d(xd, wd, bd, stride=stride, pad=pad, pad_mode=pad_mode, act=act)
    else:
        y = ops.conv_transpose2d(xd, wd, bd, stride=stride, pad=pad, out_pad=1, act=act)
    n1 = lib.mt_kernel_variant_launches(4)
    gy = torch.randn(*y.shape, generator=g).bfloat16().float()
    y.backward(gy.to(dev))
    n2 = lib.mt_kernel_variant_launches(4)
    return (x, w, b, gy), (y.detach().float().cpu(), xd.grad.detach().float().cpu(), wd.grad.detach().float().cpu()), (n1 - n0, n2 - n1)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_wsreg_matches_reference_and_replaced_kernels(case, hip_device):
    from masterthesis_amd import _lib as L, hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    lib = L.load()
    prev = lib.mt_kernel_variant_enable(4, 1)
    try:
        (x, w, b, gy), (y1, dx1, dw1), used = _run(ops, lib, case, hip_device)
        want = case[13]
        assert used[0] == want[0], f"weight-stationary launches in the forward = {used[0]}, expected {want[0]}"
        if want[1] is not None:
            assert used[1] == want[1], f"weight-stationary launches in the backward = {used[1]}, expected {want[1]}"
        lib.mt_kernel_variant_enable(4, 0)
        _, (y0, dx0, dw0), unused = _run(ops, lib, case, hip_device)
        assert unused == (0, 0)
    finally:
        lib.mt_kernel_variant_enable(4, prev)
    check_against_reference(case[:13], x, w, b, gy, y1, dx1)
    for new, old, what in ((y1, y0, "fwd"), (dx1, dx0, "dx"), (dw1, dw0, "dw")):
        rel = (new - old).norm().item() / (old.norm().item() + 1e-12)
        assert rel < 4e-3, f"{case[0]} {what}: rel L2 {rel:.3e} between the weight-stationary and the replaced kernel"
        assert torch.isfinite(new).all()


@pytest.mark.parametrize("shape", [(4, 64, 256, 256, 128, 2), (6, 64, 128, 128, 64, 1)], ids=["ec_down_s2", "s1_64_64"])
def test_wsreg_fused_statistics(shape, hip_device):
    """conv2d(..., stats=True): per-(image, channel) {sum, sum of squares} of the output from the kernel's registers (flushed once
    per image and wave) against the sums of the output it wrote, and the output against the tile kernel."""
    from masterthesis_amd import _lib as L, hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    lib = L.load()
    N, Ci, H, W, Co, stride = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().to(hip_device)
    w = (torch.randn(Co, Ci, 3, 3, generator=g) * (Ci * 9) ** -0.5).bfloat16().float().to(hip_device)
    b = (torch.randn(Co, generator=g) * 0.1).to(hip_device)
    prev = lib.mt_kernel_variant_enable(4, 1)
    try:
        n0 = lib.mt_kernel_variant_launches(4)
        with torch.no_grad():
            y, sums = ops.conv2d(x, w, b, stride=stride, pad=1, pad_mode="reflect", stats=True)
        assert lib.mt_kernel_variant_launches(4) == n0 + 1 and sums is not None
        lib.mt_kernel_variant_enable(4, 0)
        with torch.no_grad():
            y0, sums0 = ops.conv2d(x, w, b, stride=stride, pad=1, pad_mode="reflect", stats=True)
    finally:
        lib.mt_kernel_variant_enable(4, prev)
    torch.cuda.synchronize()
    yf = ops.to_nchw_f32(y)
    assert ((yf - ops.to_nchw_f32(y0)).norm() / yf.norm()).item() < 4e-3
    # the statistics are taken of the fp32 values BEFORE the bf16 rounding of the store: compare with the tile kernel's (same rule)
    s = sums[:, :Co].cpu()
    s0 = sums0[:, :Co].cpu()
    assert torch.allclose(s[..., 0], s0[..., 0], rtol=1e-3, atol=1e-3 * s0[..., 0].abs().max().item())
    assert torch.allclose(s[..., 1], s0[..., 1], rtol=1e-3, atol=1e-3 * s0[..., 1].abs().max().item())
    ref1 = yf.double().sum((2, 3)).float().cpu()
    assert torch.allclose(s[..., 0], ref1, rtol=2e-2, atol=2e-2 * ref1.abs().max().item())
