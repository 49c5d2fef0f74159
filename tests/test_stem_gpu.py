"""The direct 7x7 stem kernel (csrc/stem_kernel.hip) against the gather-GEMM it replaces (same values up to the fp32
summation order) and against the fp32 CPU reference of the op: ragged tiles, both padding modes, bias, activation, and
the fused InstanceNorm statistics."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # name, N, Ci, H, W, pad_mode, bias, act
    ("reflect_bias", 2, 3, 64, 64, "reflect", True, None),
    ("ragged", 3, 3, 40, 53, "reflect", True, None),
    ("zero_relu", 2, 3, 33, 47, "zero", False, "relu"),
    ("four_channels_lrelu", 1, 4, 48, 32, "reflect", True, "lrelu"),
    ("bench_size", 4, 3, 256, 256, "reflect", True, None),
]


def _variant(lib, on):
    return lib.mt_kernel_variant_enable(1, on)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_stem_direct_matches(case, hip_device):
    from masterthesis_amd import hip_ops as ops, _lib
    ops.set_compute_dtype(torch.bfloat16)
    lib = _lib.load()
    name, N, Ci, H, W, pad_mode, bias, act = case
    g = torch.Generator().manual_seed(len(name))
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(64, Ci, 7, 7, generator=g) * (Ci * 49) ** -0.5).bfloat16().float()
    b = torch.randn(64, generator=g) * 0.1 if bias else None
    xd, wd = x.to(hip_device), w.to(hip_device)
    bd = b.to(hip_device) if bias else None
    outs = []
    prev = _variant(lib, 1)
    try:
        for on in (1, 0):
            _variant(lib, on)
            n0 = lib.mt_kernel_variant_launches(1)
            with torch.no_grad():
                y = ops.conv2d(xd, wd, bd, stride=1, pad=3, pad_mode=pad_mode, act=act)
            used = lib.mt_kernel_variant_launches(1) - n0
            assert used == (1 if on else 0), (on, used)
            outs.append(y.float().cpu())
    finally:
        _variant(lib, prev)
    xp = F.pad(x, (3,) * 4, mode="reflect") if pad_mode == "reflect" else F.pad(x, (3,) * 4)
    ref = F.conv2d(xp, w, b)
    if act == "relu":
        ref = F.relu(ref)
    elif act == "lrelu":
        ref = F.leaky_relu(ref, 0.01)
    for got, what in ((outs[0], "direct"), (outs[1], "gather-GEMM")):
        rel = (got - ref).norm().item() / ref.norm().item()
        assert rel < 5e-3, f"{what}: rel L2 err {rel:.3e}"
        assert (got - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-2, what
    # the two kernels differ only by fp32 summation order: at most a bf16 ulp apart, and almost everywhere identical
    d = (outs[0] - outs[1]).abs()
    assert d.max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-3
    assert (d > 0).float().mean().item() < 0.2


@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 48, 80)])
def test_stem_direct_statistics(shape, hip_device):
    """conv + fused statistics -> InstanceNorm: the normalised output with the direct kernel equals the gather-GEMM's"""
    from masterthesis_amd import hip_ops as ops, _lib
    ops.set_compute_dtype(torch.bfloat16)
    lib = _lib.load()
    N, H, W = shape
    g = torch.Generator().manual_seed(H)
    x = torch.randn(N, 3, H, W, generator=g).bfloat16().float().to(hip_device)
    w = (torch.randn(64, 3, 7, 7, generator=g) * 147 ** -0.5).bfloat16().float().to(hip_device)
    b = (torch.randn(64, generator=g) * 0.1).to(hip_device)
    res = []
    prev = _variant(lib, 1)
    try:
        for on in (1, 0):
            _variant(lib, on)
            n0 = lib.mt_kernel_variant_launches(1)
            with torch.no_grad():
                y, sums = ops.conv2d(x, w, b, stride=1, pad=3, pad_mode="reflect", stats=True)
            if on:
                assert lib.mt_kernel_variant_launches(1) == n0 + 1
            yf = y.float()
            if sums is None:       # (a shape without the fused epilogue: take the statistics of the output)
                s = torch.stack([yf.sum((2, 3)), (yf * yf).sum((2, 3))], -1)
            else:
                s = sums.float()[:, :64]
            res.append((yf.cpu(), s.cpu()))
    finally:
        _variant(lib, prev)
    (y1, s1), (y0, s0) = res
    ref_s = torch.stack([y1.sum((2, 3)), (y1 * y1).sum((2, 3))], -1)
    assert torch.allclose(s1, ref_s, rtol=2e-3, atol=2e-2 * (H * W) ** 0.5), (s1 - ref_s).abs().max()
    assert torch.allclose(s1, s0, rtol=5e-3, atol=5e-2 * (H * W) ** 0.5), (s1 - s0).abs().max()


WG_CASES = [
    # name, N, Ci, H, W, pad_mode
    ("reflect", 2, 3, 64, 64, "reflect"),
    ("ragged", 3, 3, 40, 53, "reflect"),
    ("zero", 2, 3, 33, 47, "zero"),
    ("four_channels", 1, 4, 48, 32, "reflect"),
    ("bench_size", 4, 3, 256, 256, "reflect"),
]


@pytest.mark.parametrize("case", WG_CASES, ids=[c[0] for c in WG_CASES])
def test_stem_direct_weight_gradient(case, hip_device):
    """dW of the 7x7 stem: direct kernel vs the gather form vs the fp32 reference (both accumulate in fp32; the
    operands are bf16-exact, so all three agree to fp32 summation noise)"""
    from masterthesis_amd import hip_ops as ops, _lib
    ops.set_compute_dtype(torch.bfloat16)
    lib = _lib.load()
    name, N, Ci, H, W, pad_mode = case
    g = torch.Generator().manual_seed(7 + len(name))
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(64, Ci, 7, 7, generator=g) * (Ci * 49) ** -0.5).bfloat16().float()
    gy = torch.randn(N, 64, H, W, generator=g).bfloat16().float()
    xr, wr = x.clone(), w.clone().requires_grad_()
    xp = F.pad(xr, (3,) * 4, mode="reflect") if pad_mode == "reflect" else F.pad(xr, (3,) * 4)
    F.conv2d(xp, wr).backward(gy)
    grads = []
    prev = lib.mt_kernel_variant_enable(1, 1)
    try:
        for on in (1, 0):
            lib.mt_kernel_variant_enable(1, on)
            wd = w.to(hip_device).requires_grad_()
            y = ops.conv2d(x.to(hip_device), wd, None, stride=1, pad=3, pad_mode=pad_mode)
            n0 = lib.mt_kernel_variant_launches(1)
            y.backward(gy.to(hip_device))
            if on:
                assert lib.mt_kernel_variant_launches(1) == n0 + 1, "the weight gradient was meant to run on the direct kernel"
            grads.append(wd.grad.float().cpu())
            # accumulate into an existing gradient (the fused in-place path of the training step)
            y2 = ops.conv2d(x.to(hip_device), wd, None, stride=1, pad=3, pad_mode=pad_mode)
            y2.backward(gy.to(hip_device))
            assert torch.allclose(wd.grad.float().cpu(), 2 * grads[-1], rtol=1e-4, atol=1e-3)
    finally:
        lib.mt_kernel_variant_enable(1, prev)
    ref = wr.grad
    scale = ref.abs().max().item()
    for got, what in ((grads[0], "direct"), (grads[1], "gather form")):
        assert (got - ref).abs().max().item() <= 2e-4 * scale + 1e-3, (what, (got - ref).abs().max().item(), scale)


@pytest.mark.parametrize("case", WG_CASES, ids=[c[0] for c in WG_CASES])
def test_stem_direct_data_gradient(case, hip_device):
    """dx of the 7x7 stem: direct kernel (+ reflect fold) vs the gather form vs the fp32 reference"""
    from masterthesis_amd import hip_ops as ops, _lib
    ops.set_compute_dtype(torch.bfloat16)
    lib = _lib.load()
    name, N, Ci, H, W, pad_mode = case
    g = torch.Generator().manual_seed(11 + len(name))
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(64, Ci, 7, 7, generator=g) * (Ci * 49) ** -0.5).bfloat16().float()
    gy = torch.randn(N, 64, H, W, generator=g).bfloat16().float()
    xr = x.clone().requires_grad_()
    xp = F.pad(xr, (3,) * 4, mode="reflect") if pad_mode == "reflect" else F.pad(xr, (3,) * 4)
    F.conv2d(xp, w).backward(gy)
    grads = []
    prev = lib.mt_kernel_variant_enable(1, 1)
    try:
        for on in (1, 0):
            lib.mt_kernel_variant_enable(1, on)
            xd = x.to(hip_device).requires_grad_()
            y = ops.conv2d(xd, w.to(hip_device), None, stride=1, pad=3, pad_mode=pad_mode)
            n0 = lib.mt_kernel_variant_launches(1)
            y.backward(gy.to(hip_device))
            if on:
                assert lib.mt_kernel_variant_launches(1) == n0 + 1, "the data gradient was meant to run on the direct kernel"
            grads.append(xd.grad.float().cpu())
    finally:
        lib.mt_kernel_variant_enable(1, prev)
    ref = xr.grad
    for got, what in ((grads[0], "direct"), (grads[1], "gather form")):
        rel = (got - ref).norm().item() / ref.norm().item()
        assert rel < 5e-3, f"{what}: rel L2 err {rel:.3e}"
        assert (got - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item() + 1e-2, what
    d = (grads[0] - grads[1]).abs()
    assert d.max().item() <= 2 ** -6 * ref.abs().max().item() + 1e-3
