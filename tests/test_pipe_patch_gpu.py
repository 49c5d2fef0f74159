"""The 256x256 ping-pong gather-GEMM with a patch-resident pixel operand (csrc/conv_pipe_patch_kernel.hip) against the
ring kernel it replaces (csrc/conv_pipe_kernel.hip) -- BIT FOR BIT: both walk K slice-major with the taps of a slice
back to back and accumulate the same MFMA fragments in the same order; only the way the pixel fragment reaches LDS
differs (one patch copy per slice instead of one tile copy per tap) -- and against the fp32 CPU reference.

Shapes: the residual-block convolution of the step (3x3 256->256 on 64x64, reflection padding: forward, forward with
fused statistics, interior of the data gradient), the same on 32x32 / 16x16 / 128x128 maps (8, 16, 2 tile rows), two
weight tiles (512 output channels), 5x5 taps, zero padding, fp32 storage.  Where dx has a multiple of 256 channels the data
gradient of the reflection-padded cases runs as ONE launch with the fold inside the pixel operand: equal to the
interior + ring + fold path bit for bit away from the rows / columns 1 and H-2 / W-2, to bf16 rounding on them."""
import pytest
import torch
import torch.nn.functional as F

from test_persist_gpu import case_seed, check_against_reference

pytestmark = pytest.mark.gpu

# name, N, Ci, H, W, Co, k, pad, pad_mode, act, dtype
CASES = [
    ("k1_64x64", 16, 256, 64, 64, 256, 3, 1, "reflect", None, "bf16"),
    ("k1_64x64_relu_N32", 32, 256, 64, 64, 256, 3, 1, "reflect", "relu", "bf16"),
    ("k1_32x32_N64", 64, 256, 32, 32, 256, 3, 1, "reflect", None, "bf16"),
    ("k1_16x16_zero", 256, 64, 16, 16, 256, 3, 1, "zero", "lrelu", "bf16"),
    ("k1_128x128_co512", 6, 32, 128, 128, 512, 3, 1, "reflect", None, "bf16"),
    ("k5_64x64", 16, 64, 64, 64, 256, 5, 2, "reflect", None, "bf16"),
    ("k1_64x64_fp32", 16, 64, 64, 64, 256, 3, 1, "reflect", None, "fp32"),
    ("k1_32x32_fp32_fold", 64, 256, 32, 32, 256, 3, 1, "reflect", None, "fp32"),
]


def _run(ops, lib, case, dev, stats):
    name, N, Ci, H, W, Co, k, pad, pad_mode, act, dtype = case
    g = torch.Generator().manual_seed(case_seed(name))
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, k, k, generator=g) * (Ci * k * k) ** -0.5).bfloat16().float()
    xd = x.to(dev).requires_grad_()
    wd = w.to(dev).requires_grad_()
    n0 = lib.mt_kernel_variant_launches(3)
    out = ops.conv2d(xd, wd, None, stride=1, pad=pad, pad_mode=pad_mode, act=act, stats=stats)
    y, sums = out if stats else (out, None)
    n1 = lib.mt_kernel_variant_launches(3)
    gy = torch.randn(*y.shape, generator=g).bfloat16().float()
    y.backward(gy.to(dev))
    n2 = lib.mt_kernel_variant_launches(3)
    return (x, w, gy), (y.detach().float().cpu(), xd.grad.detach().float().cpu(),
                        None if sums is None else sums.detach().float().cpu()), (n1 - n0, n2 - n1)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_patch_resident_ping_pong_is_bit_identical_to_the_ring_kernel(case, hip_device):
    from masterthesis_amd import _lib as L, hip_ops as ops
    dtype = torch.bfloat16 if case[10] == "bf16" else torch.float32
    ops.set_compute_dtype(dtype)
    lib = L.load()
    stats = case[9] is None and dtype == torch.bfloat16          # the fused-statistics epilogue needs act == none
    prev = lib.mt_kernel_variant_enable(3, 1)
    try:
        (x, w, gy), (y1, dx1, s1), used = _run(ops, lib, case, hip_device, stats)
        # forward on the new kernel; the data gradient too where dx has a multiple of 256 channels (its interior launch)
        assert used[0] == 1, f"forward launches on the patch-resident kernel: {used[0]}"
        assert used[1] == (1 if (case[2] % 256 == 0 and case[8] == "reflect") or (case[2] % 256 == 0 and case[8] == "zero") else 0), used
        lib.mt_kernel_variant_enable(3, 0)
        _, (y0, dx0, s0), unused = _run(ops, lib, case, hip_device, stats)
        assert unused == (0, 0)
    finally:
        lib.mt_kernel_variant_enable(3, prev)
        ops.set_compute_dtype(torch.bfloat16)
    assert torch.equal(y1, y0), f"forward differs from the ring kernel: max {(y1 - y0).abs().max().item():.3e}"
    if used[1] and case[8] == "reflect":
        # the data gradient ran as ONE launch with the reflection fold inside the pixel operand (the summed operand is
        # rounded to bf16 once more on the border-adjacent pixels) instead of interior + ring GEMM + fold kernel
        rel = ((dx1 - dx0).norm() / dx0.norm()).item()
        assert rel < (2e-3 if dtype == torch.bfloat16 else 1e-6), f"data gradient: rel L2 {rel:.3e} to the interior + ring + fold path"
        band = torch.zeros_like(dx1, dtype=torch.bool)
        band[:, :, 1], band[:, :, -2], band[:, :, :, 1], band[:, :, :, -2] = True, True, True, True
        assert torch.equal(dx1[~band], dx0[~band]), "pixels that receive no reflected contribution must be bit-identical"
    else:
        assert torch.equal(dx1, dx0), f"data gradient differs from the ring kernel: max {(dx1 - dx0).abs().max().item():.3e}"
    if stats:
        # (the statistics are accumulated with float atomics: equal up to their order)
        assert torch.allclose(s1, s0, rtol=1e-4, atol=1e-3), (s1 - s0).abs().max()
    if 2.0 * case[1] * case[3] * case[4] * case[2] * case[5] * case[6] ** 2 > 1e11:
        return          # (a CPU reference of this size takes minutes; the ring kernel it equals bit for bit is checked at this size
                        #  through the adjoint identities of tests/test_fullsize_gpu.py)
    if dtype == torch.bfloat16:
        name, N, Ci, H, W, Co, k, pad, pad_mode, act, _ = case
        check_against_reference((name, "conv", N, Ci, H, W, Co, k, 1, pad, pad_mode, False, act), x, w, None, gy, y1, dx1)
    else:
        xr = x.clone().requires_grad_()
        yr = F.conv2d(F.pad(xr, (case[7],) * 4, mode="reflect"), w)
        yr.backward(gy)
        assert ((y1 - yr.detach()).norm() / yr.detach().norm()).item() < 1e-5
        assert ((dx1 - xr.grad).norm() / xr.grad.norm()).item() < 1e-5


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
def test_skip_gradient_rides_on_the_data_gradient_epilogue(dtype, hip_device):
    """ops.GradLink: the skip-connection gradient parked by the residual norm's backward is added inside the epilogue of
    conv1's data-gradient GEMM (mt_conv_bwd_data_add on the patch-resident kernel) -- dx must be dgrad + skip, to one
    rounding of the sum (bf16) / to summation order (fp32) -- and through the library's add where the kernel does not apply."""
    from masterthesis_amd import _lib as L, hip_ops as ops
    ops.set_compute_dtype(dtype)
    lib = L.load()
    try:
        for (N, C, H) in ((16, 256, 64), (2, 24, 20)):          # fused epilogue / fallback add
            g = torch.Generator().manual_seed(N + C)
            x = torch.randn(N, C, H, H, generator=g).bfloat16().float().to(hip_device)
            w = (torch.randn(C, C, 3, 3, generator=g) * (C * 9) ** -0.5).bfloat16().float().to(hip_device)
            gy = ops.canon(torch.randn(N, C, H, H, generator=g).bfloat16().float().to(hip_device))
            skip = ops.canon(torch.randn(N, C, H, H, generator=g).bfloat16().float().to(hip_device))
            xa = x.clone().requires_grad_()
            ops.conv2d(xa, w, None, stride=1, pad=1, pad_mode="reflect").backward(gy)
            link = ops.GradLink()
            xb = x.clone().requires_grad_()
            n0 = lib.mt_kernel_variant_launches(3)
            y = ops.conv2d(xb, w, None, stride=1, pad=1, pad_mode="reflect", grad_link=link)
            link.g = skip
            y.backward(gy)
            used = lib.mt_kernel_variant_launches(3) - n0
            assert link.g is None and used == (2 if C == 256 else 0)
            want = ops.to_nchw_f32(xa.grad) + ops.to_nchw_f32(skip)
            got = ops.to_nchw_f32(xb.grad)
            tol = 2 ** -8 if dtype == torch.bfloat16 else 1e-5
            assert (got - want).abs().max().item() <= tol * want.abs().max().item() + 1e-6
    finally:
        ops.set_compute_dtype(torch.bfloat16)


# (fp32: the two runs differ in the order of EVERY statistics reduction, forward ones included -- 3.7e-4 measured through the
#  six normalisation backwards, whose mean subtraction amplifies round-off; a wrong mask or a lost skip gradient is O(0.1-1))
# (bf16: 3.8e-2 measured between the two modes -- bf16 rounding of six layers of activations and gradients under two
#  different summation orders, with ReLU masks in between)
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 8e-2)], ids=["fp32", "bf16"])
def test_norm_backward_statistics_from_the_data_gradient_epilogue(dtype, tol, hip_device):
    """A chain [conv-IN-ReLU] -> ResnetBlock -> ResnetBlock at the step's size (256 channels, 64x64, 16 images): with the
    gradient / statistics links the skip gradients are added and the sums of five of the six normalisation backwards are
    taken inside the data-gradient epilogues (ops.GradLink / ops.StatsLink, mt_conv_bwd_data_ex); deterministic mode runs
    every statistics pass on its own.  Same gradients either way."""
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd.models.core import blocks as B
    from masterthesis_amd.models.core.functions import init_weights
    import torch.nn as nn
    ops.set_compute_dtype(dtype)
    try:
        torch.manual_seed(1)
        net = nn.Sequential(B.ConvBlock(256, 256, 3, 1, 1, padding_type="reflect", norm_layer="instance", activation="relu"),
                            B.ResnetBlock(256, 256), B.ResnetBlock(256, 256))
        init_weights(net, "normal", 0.05)
        net = net.to(hip_device)
        x0 = torch.randn(16, 256, 64, 64, generator=torch.Generator().manual_seed(2)).to(hip_device)
        gy = ops.canon(torch.randn(16, 256, 64, 64, generator=torch.Generator().manual_seed(3)).to(hip_device))
        res = {}
        ops.set_stats_link(True)
        for det in (True, False):
            ops.set_deterministic(det)
            for p in net.parameters():
                p.grad = None
            x = x0.clone().requires_grad_()
            y = net(x)
            ops.hbm_timer_start()
            y.backward(gy)
            used = ops.hbm_timer_stop()
            # normalisation backwards that made their own pass for the statistics: the three-launch form's mt_nc_stats_bwd, or
            # (bf16) the one-pass kernel, which holds the statistics pass inside
            calls = used.get("nc_stats_bwd", (0, 0, 0))[0] + used.get("norm_bwd_onepass", (0, 0, 0))[0]
            assert ("norm_bwd_onepass" in used) == (dtype == torch.bfloat16)
            res[det] = (ops.to_nchw_f32(x.grad).cpu(), [p.grad.detach().cpu().clone() for p in net.parameters()], calls)
        assert res[True][2] == 5 and res[False][2] == 1, (res[True][2], res[False][2])     # only the last norm keeps its own pass
        a, b = res[False], res[True]
        assert ((a[0] - b[0]).norm() / b[0].norm()).item() < tol
        for ga, gb in zip(a[1], b[1]):
            if gb.norm().item() > 1e-6 * max(g.norm().item() for g in b[1]):
                assert ((ga - gb).norm() / gb.norm()).item() < tol
    finally:
        ops.set_deterministic(False)
        ops.set_stats_link(False)
        ops.set_compute_dtype(torch.bfloat16)
