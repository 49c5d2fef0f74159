"""The patch-resident gather-GEMM (csrc/conv_patch_kernel.hip: the pixel tile's input patch stays in LDS and serves all
taps and all four sub-pixel phases) against the fp32 CPU reference of the op, and against the kernels it replaces.

Every case asserts that the launches it is about really ran on the new kernel (mt_kernel_variant_launches(2)): stride-1
forward (gather form, one phase: 128- and 64-channel tiles, several channel tiles, reflection and zero padding, ragged
tiles, 16- and 32-wide tiles), transposed convolution forward and strided data gradient (scatter form, four phases
from one patch; with reflection padding the interior goes straight into dx and only the ring into the workspace),
4x4 stride-2 (nine input offsets, sixteen taps)."""
import pytest
import torch
import torch.nn.functional as F

from test_persist_gpu import case_seed, check_against_reference

pytestmark = pytest.mark.gpu

# name, kind, N, Ci, H, W, Co, k, stride, pad, pad_mode, bias, act, (patch launches expected in fwd, in bwd)
CASES = [
    ("g128_k3s1_64_128_reflect", "conv", 4, 64, 128, 128, 128, 3, 1, 1, "reflect", True, "relu", (1, 1)),
    ("g128_two_channel_tiles", "conv", 12, 128, 64, 64, 256, 3, 1, 1, "zero", False, "lrelu", (1, 1)),
    ("g64_k3s1_64_64", "conv", 6, 64, 128, 128, 64, 3, 1, 1, "reflect", True, None, (1, 1)),
    ("g64_ragged_96_192", "conv", 3, 96, 60, 90, 192, 3, 1, 1, "zero", True, "lrelu", (1, 0)),      # (dx has 96 channels: not a multiple of 64)
    ("g128_tw16_48x48", "conv", 24, 128, 48, 48, 128, 3, 1, 1, "reflect", False, None, (1, 1)),
    ("s_convT_128_64", "convT", 6, 128, 64, 64, 64, 3, 2, 1, "zero", True, None, (1, 0)),
    ("s_convT_256_128_relu", "convT", 12, 256, 32, 32, 128, 3, 2, 1, "zero", True, "relu", (1, 0)),
    ("s_dgrad_k3s2_reflect", "conv", 6, 64, 128, 128, 128, 3, 2, 1, "reflect", True, None, (0, 1)),
    ("s_dgrad_k4s2_zero", "conv", 24, 64, 64, 64, 128, 4, 2, 1, "zero", False, "lrelu", (0, 1)),
    ("s_dgrad_k3s2_odd", "conv", 8, 128, 63, 65, 64, 3, 2, 1, "zero", True, None, (0, 1)),
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
    n0 = lib.mt_kernel_variant_launches(2)
    if kind == "conv":
        y = ops.conv2d(xd, wd, bd, stride=stride, pad=pad, pad_mode=pad_mode, act=act)
    else:
        y = ops.conv_transpose2d(xd, wd, bd, stride=stride, pad=pad, out_pad=1, act=act)
    n1 = lib.mt_kernel_variant_launches(2)
    gy = torch.randn(*y.shape, generator=g).bfloat16().float()
    y.backward(gy.to(dev))
    n2 = lib.mt_kernel_variant_launches(2)
    return (x, w, b, gy), (y.detach().float().cpu(), xd.grad.detach().float().cpu()), (n1 - n0, n2 - n1)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_patch_kernel_matches_reference_and_replaced_kernels(case, hip_device):
    from masterthesis_amd import _lib as L, hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    lib = L.load()
    prev = lib.mt_kernel_variant_enable(2, 2)       # every shape the kernel can run (the default takes only the measured wins)
    prev_ws = lib.mt_kernel_variant_enable(4, 0)    # (round 4: the weight-stationary kernel would take the 64-channel shapes first)
    try:
        (x, w, b, gy), (y1, dx1), used = _run(ops, lib, case, hip_device)
        assert used == case[13], f"patch-kernel launches (forward, backward) = {used}, expected {case[13]}"
        lib.mt_kernel_variant_enable(2, 0)
        _, (y0, dx0), unused = _run(ops, lib, case, hip_device)
        assert unused == (0, 0)
    finally:
        lib.mt_kernel_variant_enable(2, prev)
        lib.mt_kernel_variant_enable(4, prev_ws)
    # against the fp32 reference of the op (element-wise through the device's own activation mask)
    check_against_reference(case[:13], x, w, b, gy, y1, dx1)
    # against the kernels it replaces: same bf16 operands, fp32 accumulation in another order, bf16 outputs
    for new, old, what in ((y1, y0, "fwd"), (dx1, dx0, "dx")):
        rel = (new - old).norm().item() / (old.norm().item() + 1e-12)
        assert rel < 4e-3, f"{case[0]} {what}: rel L2 {rel:.3e} between the patch-resident and the replaced kernel"
