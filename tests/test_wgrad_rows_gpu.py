"""The accumulator-stationary weight gradient of round 4 (csrc/wgrad_rows_kernel.hip: a workgroup keeps a 64 x 9 x 64 block of dW
in registers, one wave per filter tap, and walks down a 32-pixel column strip with the input rows in an LDS ring) against the fp32
CPU reference of the op and against the 128 x 128 tile kernel it replaces.  Shapes are the 3x3 layers between the stem and the
256-channel bottleneck (reference networks.py:33,248; blocks.py:73,93-119): stride 1 and stride 2, reflection / zero padding, the
transposed convolutions (operand roles swapped), several channel blocks per operand, a row split that does not divide the map
height.  Every case asserts that its weight gradient really ran on the new kernel (mt_kernel_variant_launches(5))."""
import pytest
import torch
import torch.nn.functional as F

from test_persist_gpu import case_seed

pytestmark = pytest.mark.gpu

# name, kind, N, Ci, H, W, Co, stride, pad_mode
CASES = [
    ("s1_64_64_reflect", "conv", 8, 64, 128, 128, 64, 1, "reflect"),
    ("s1_64_128_reflect", "conv", 4, 64, 128, 128, 128, 1, "reflect"),
    ("s1_128_128_zero", "conv", 8, 128, 64, 64, 128, 1, "zero"),
    ("s1_128_256_reflect", "conv", 8, 128, 64, 64, 256, 1, "reflect"),
    ("s1_64_64_zero_ragged_rows", "conv", 24, 64, 104, 64, 64, 1, "zero"),
    ("s1_64_64_reflect_ragged_rows", "conv", 24, 64, 104, 64, 64, 1, "reflect"),
    ("s2_64_128_reflect", "conv", 16, 64, 128, 128, 128, 2, "reflect"),
    ("s2_128_256_zero", "conv", 16, 128, 64, 64, 256, 2, "zero"),
    ("s2_64_128_zero_wide", "conv", 8, 64, 128, 256, 128, 2, "zero"),
    ("convT_128_64", "convT", 16, 128, 64, 64, 64, 2, "zero"),
    ("convT_256_128", "convT", 16, 256, 32, 32, 128, 2, "zero"),
]


def _reference(case, x, w, gy):
    name, kind, N, Ci, H, W, Co, stride, pad_mode = case
    wr = w.clone().requires_grad_()
    if kind == "conv":
        xp = F.pad(x, (1, 1, 1, 1), mode="reflect") if pad_mode == "reflect" else F.pad(x, (1, 1, 1, 1))
        y = F.conv2d(xp, wr, None, stride=stride)
    else:
        y = F.conv_transpose2d(x, wr, None, stride=stride, padding=1, output_padding=1)
    y.backward(gy)
    return wr.grad


def _run(ops, lib, case, dev):
    name, kind, N, Ci, H, W, Co, stride, pad_mode = case
    g = torch.Generator().manual_seed(case_seed(name))
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    wshape = (Co, Ci, 3, 3) if kind == "conv" else (Ci, Co, 3, 3)
    w = (torch.randn(*wshape, generator=g) * (Ci * 9) ** -0.5).bfloat16().float()
    xd = x.to(dev)
    wd = w.to(dev).requires_grad_()
    if kind == "conv":
        y = ops.conv2d(xd, wd, None, stride=stride, pad=1, pad_mode=pad_mode)
    else:
        y = ops.conv_transpose2d(xd, wd, None, stride=stride, pad=1, out_pad=1)
    gy = torch.randn(*y.shape, generator=g).bfloat16().float()
    n0 = lib.mt_kernel_variant_launches(5)
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    n1 = lib.mt_kernel_variant_launches(5)
    return (x, w, gy), wd.grad.detach().float().cpu(), n1 - n0


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_row_walker_weight_gradient_matches_reference_and_tile_kernel(case, hip_device):
    from masterthesis_amd import _lib as L, hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    lib = L.load()
    prev = lib.mt_kernel_variant_enable(5, 1)
    try:
        (x, w, gy), dw1, used = _run(ops, lib, case, hip_device)
        assert used == 1, f"weight-gradient launches on the row walker: {used}"
        lib.mt_kernel_variant_enable(5, 0)
        _, dw0, unused = _run(ops, lib, case, hip_device)
        assert unused == 0
    finally:
        lib.mt_kernel_variant_enable(5, prev)
    ref = _reference(case, x, w, gy)
    # bf16 operands, fp32 products and sums on both sides: only the summation order differs
    rel = ((dw1 - ref).norm() / ref.norm()).item()
    assert rel < 2e-5, f"{case[0]}: rel L2 {rel:.3e} to the fp32 CPU reference"
    worst = ((dw1 - ref).abs().max() / ref.abs().max()).item()
    assert worst < 2e-4, f"{case[0]}: max error {worst:.3e} of the largest gradient element"
    rel0 = ((dw1 - dw0).norm() / dw0.norm()).item()
    assert rel0 < 2e-5, f"{case[0]}: rel L2 {rel0:.3e} to the tile kernel"


def test_row_walker_declines_what_it_cannot_walk(hip_device):
    """4x4 windows, maps narrower than a 32-pixel strip and channel counts off the 64-wide blocks keep the tile kernels"""
    from masterthesis_amd import _lib as L, hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    lib = L.load()
    dev = hip_device
    for (N, Ci, H, W, Co, k, s, p) in [(16, 64, 64, 64, 128, 4, 2, 1), (64, 256, 16, 16, 256, 3, 1, 1), (16, 48, 64, 64, 64, 3, 1, 1)]:
        x = torch.randn(N, Ci, H, W, device=dev)
        w = torch.randn(Co, Ci, k, k, device=dev, requires_grad=True)
        n0 = lib.mt_kernel_variant_launches(5)
        ops.conv2d(x, w, None, stride=s, pad=p).sum().backward()
        assert lib.mt_kernel_variant_launches(5) == n0
        assert torch.isfinite(w.grad).all()


def _net_grads(ops, dev, seed, N):
    """an encoder / decoder shaped chain of eligible layers, one of them applied to two inputs (two uses of one weight)"""
    g = torch.Generator().manual_seed(seed)

    def w_(*shape):
        fan = shape[1] * 9
        w = (torch.randn(*shape, generator=g) * fan ** -0.5).bfloat16().float().to(dev).requires_grad_()
        w.grad = torch.zeros_like(w)               # (accumulated into in place, as the optimizer's flat gradient buffers are)
        return w
    ws = {"down1": w_(128, 64, 3, 3), "same": w_(128, 128, 3, 3), "down2": w_(256, 128, 3, 3), "up1": w_(256, 128, 3, 3),
          "up2": w_(128, 64, 3, 3), "s64": w_(64, 64, 3, 3), "res": w_(256, 256, 3, 3)}
    # a bias whose gradient rides on the weight-gradient call (no fused activation behind the transposed convolution)
    ws["up2_bias"] = (torch.randn(64, generator=g) * 0.1).to(dev).requires_grad_()
    ws["up2_bias"].grad = torch.zeros_like(ws["up2_bias"])
    xa = torch.randn(N, 64, 128, 128, generator=g).bfloat16().float().to(dev)
    xb = torch.randn(N, 64, 128, 128, generator=g).bfloat16().float().to(dev)

    def enc(x):
        h = ops.conv2d(x, ws["s64"], None, stride=1, pad=1, pad_mode="reflect", act="relu")
        h = ops.conv2d(h, ws["down1"], None, stride=2, pad=1, pad_mode="reflect", act="relu")
        h = ops.conv2d(h, ws["same"], None, stride=1, pad=1, pad_mode="zero", act="lrelu")
        return ops.conv2d(h, ws["down2"], None, stride=2, pad=1, pad_mode="reflect")
    ha, hb = enc(xa), enc(xb)                      # every encoder weight is used twice in this backward pass
    # (a 256 -> 256 layer on a 32 x 32 map: a ping-pong shape that joins the shared launches because its map is small)
    hr = ops.conv2d(ha + hb, ws["res"], None, stride=1, pad=1, pad_mode="reflect", act="relu")
    h = ops.conv_transpose2d(hr, ws["up1"], None, stride=2, pad=1, out_pad=1, act="relu")
    y = ops.conv_transpose2d(h, ws["up2"], ws["up2_bias"], stride=2, pad=1, out_pad=1)
    t = torch.randn(*y.shape, generator=g).to(dev)
    (y.float() * t).sum().backward()
    torch.cuda.synchronize()
    return {k: v.grad.detach().float().cpu() for k, v in ws.items()}


def test_row_walker_shared_launches_match_per_layer_tile_kernels(hip_device):
    """eleven weight-gradient problems of one backward pass (seven weights, four of them used twice) parked and handed over at once:
    two launches of the row walker (stride 1 / stride 2 class) and one batched slab sum -- against the same pass with every
    weight gradient on the tile kernels where autograd reaches it"""
    from masterthesis_amd import _lib as L, hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    lib = L.load()
    prev = lib.mt_kernel_variant_enable(5, 1)
    fused_before = ops._FUSE_WGRAD_ACC[0]
    try:
        ops.set_fused_grad_accumulation(True)          # (what the models' FusedAdam switches on: gradients added into param.grad)
        ops.set_wgrad_rows_multi(True)
        n0 = lib.mt_kernel_variant_launches(5)
        g1 = _net_grads(ops, hip_device, 20261005, 4)
        used = lib.mt_kernel_variant_launches(5) - n0
        assert used == 2, f"launches of the row walker for the whole pass: {used}"
        ops.set_wgrad_rows_multi(False)
        lib.mt_kernel_variant_enable(5, 0)
        n0 = lib.mt_kernel_variant_launches(5)
        g0 = _net_grads(ops, hip_device, 20261005, 4)
        assert lib.mt_kernel_variant_launches(5) == n0
    finally:
        ops.set_wgrad_rows_multi(True)
        ops.set_fused_grad_accumulation(fused_before)
        lib.mt_kernel_variant_enable(5, prev)
    for k in g0:
        rel = ((g1[k] - g0[k]).norm() / g0[k].norm()).item()
        assert rel < 2e-5, f"{k}: rel L2 {rel:.3e} between the shared launches and the per-layer tile kernels"
    assert g1["up2_bias"].abs().max() > 0


def _fuzz_cases(n=14, seed=4321):
    import random
    r = random.Random(seed)
    out = []
    for i in range(n):
        kind = r.choice(["conv", "conv", "convT"])
        stride = 2 if kind == "convT" else r.choice([1, 2])
        Ci, Co = r.choice([64, 128, 192]), r.choice([64, 128])
        if kind == "conv":
            Wo, Ho = 32 * r.randint(1, 3), r.randint(2, 41)
            H, W = Ho * stride, Wo * stride
        else:
            W, H = 32 * r.randint(1, 2), r.randint(2, 37)              # the dense operand is the INPUT of a transposed convolution
        out.append((f"fuzz{i}", kind, r.randint(1, 5), Ci, H, W, Co, stride, r.choice(["reflect", "zero"]) if kind == "conv" else "zero"))
    out += [("fuzzA", "conv", 3, 64, 14, 64, 128, 2, "reflect"), ("fuzzB", "conv", 2, 192, 42, 64, 64, 2, "reflect"),
            ("fuzzC", "conv", 1, 64, 2, 32, 64, 1, "reflect")]
    return out


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: f"{c[0]}_{c[1]}_N{c[2]}_{c[3]}x{c[4]}x{c[5]}_{c[6]}_s{c[7]}_{c[8]}")
def test_row_walker_randomised_shapes_through_the_parked_path(case, hip_device):
    """odd map heights (two rows up), one to three strips, one to five images, 64..192 channels, every padding / stride / operand
    role: the parked path takes every eligible shape whatever its size (runs of a few rows, runs that cross strips and images)"""
    from masterthesis_amd import _lib as L, hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    lib = L.load()
    name, kind, N, Ci, H, W, Co, stride, pad_mode = case
    g = torch.Generator().manual_seed(case_seed(name))
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    wshape = (Co, Ci, 3, 3) if kind == "conv" else (Ci, Co, 3, 3)
    w = (torch.randn(*wshape, generator=g) * (Ci * 9) ** -0.5).bfloat16().float()
    prev = lib.mt_kernel_variant_enable(5, 1)
    fused_before = ops._FUSE_WGRAD_ACC[0]
    try:
        ops.set_fused_grad_accumulation(True)
        ops.set_wgrad_rows_multi(True)
        wd = w.to(hip_device).requires_grad_()
        wd.grad = torch.full_like(wd, 0.5)                         # (accumulated INTO: the start value must survive)
        if kind == "conv":
            y = ops.conv2d(x.to(hip_device), wd, None, stride=stride, pad=1, pad_mode=pad_mode)
        else:
            y = ops.conv_transpose2d(x.to(hip_device), wd, None, stride=stride, pad=1, out_pad=1)
        gy = torch.randn(*y.shape, generator=g).bfloat16().float()
        n0 = lib.mt_kernel_variant_launches(5)
        (y.float() * gy.to(hip_device)).sum().backward()
        torch.cuda.synchronize()
        assert lib.mt_kernel_variant_launches(5) - n0 == 1
        got = wd.grad.detach().float().cpu() - 0.5
    finally:
        ops.set_fused_grad_accumulation(fused_before)
        lib.mt_kernel_variant_enable(5, prev)
    ref = _reference(case, x, w, gy)
    rel = ((got - ref).norm() / ref.norm()).item()
    assert rel < 3e-5, f"{case}: rel L2 {rel:.3e} to the fp32 CPU reference"
