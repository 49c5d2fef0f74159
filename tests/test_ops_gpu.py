"""GPU parity of every HIP op against a plain PyTorch fp32 CPU reference of the same op
(forward and all gradients).  fp32 path: tight tolerance; bf16 path: bf16-rounding tolerance."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: dict(rtol=2e-4, atol=2e-5), torch.bfloat16: dict(rtol=3e-2, atol=3e-2)}


def _ops(dtype):
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(dtype)
    _RND_BF16[0] = dtype == torch.bfloat16
    return ops


def _close(a, b, dtype, scale=1.0, what=""):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    tol = TOL[dtype]
    ref = b.abs().max().item() + 1e-12
    if dtype == torch.bfloat16:
        # bf16 storage: inputs/outputs carry 2^-9 relative rounding, and a ReLU mask can flip
        # where the pre-activation is within rounding of 0 -- judge by relative L2 error.
        num = (a - b).norm().item()
        den = b.norm().item() + 1e-12
        assert num <= 2e-2 * scale * den + 1e-3, f"{what}: rel L2 err {num / den:.3e}"
        return
    err = (a - b).abs().max().item()
    assert err <= tol["atol"] * scale * max(ref, 1.0) + tol["rtol"] * ref, f"{what}: max err {err:.3e}, ref max {ref:.3e}"


_RND_BF16 = [False]


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    t = torch.randn(*shape, generator=g) * scale
    # bf16 runs: make the inputs exactly representable so the fp32 reference sees the same
    # operands (otherwise ReLU masks / L1 signs flip where values are within input rounding of 0)
    return t.bfloat16().float() if _RND_BF16[0] else t


CONV_CASES = [
    # name, N, Ci, H, W, Co, k, stride, pad, pad_mode, bias, act
    ("k3s1_reflect", 2, 16, 12, 12, 16, 3, 1, 1, "reflect", False, None),
    ("k7s1_reflect_stem", 2, 3, 16, 16, 8, 7, 1, 3, "reflect", True, None),
    ("k3s2_reflect", 2, 8, 16, 16, 16, 3, 2, 1, "reflect", True, "lrelu"),
    ("k3s2_reflect_odd", 1, 8, 15, 13, 16, 3, 2, 1, "reflect", True, None),
    ("k4s2_reflect_es", 2, 5, 16, 16, 8, 4, 2, 1, "reflect", True, None),
    ("k4s2_zero_msd", 2, 8, 16, 16, 16, 4, 2, 1, "zero", False, "lrelu"),
    ("k1_pad1_patch", 2, 16, 4, 4, 1, 1, 1, 1, "zero", False, None),
    ("k4_valid_cls", 2, 16, 4, 4, 2, 4, 1, 0, "zero", False, None),
    ("k1_shortcut", 2, 8, 8, 8, 16, 1, 1, 0, "zero", True, None),
    ("k1_thin_head", 2, 128, 6, 6, 4, 1, 1, 0, "zero", True, None),
    ("k7s2_reflect_dc", 1, 8, 20, 20, 8, 7, 2, 1, "reflect", True, None),
    ("k3s1_wide", 2, 64, 20, 20, 160, 3, 1, 1, "reflect", False, "relu"),
    ("k3s1_256", 1, 256, 16, 16, 256, 3, 1, 1, "reflect", False, None),
    ("k3s2_zero_odd", 1, 24, 9, 11, 40, 3, 2, 1, "zero", True, None),
    # large enough for the 256x256-tile kernel (Cout % 256 == 0, 256 pixel tiles): forward only hits it
    ("k3s1_tile256_fwd", 4, 32, 128, 128, 256, 3, 1, 1, "reflect", True, "relu"),
    # ... and with Cin = Cout = 256 the zero-padded data gradient does too (H*W*N = 65536 pixels)
    ("k3s1_tile256_zero", 4, 256, 128, 128, 256, 3, 1, 1, "zero", False, None),
    # K1 shape class: reflect-padded data gradient = interior on the 256x256 ping-pong kernel + border ring
    ("k3s1_tile256_reflect", 16, 256, 64, 64, 256, 3, 1, 1, "reflect", False, None),
    # ring path with a 2-pixel border, non-square, and a map too small for the band enumeration of the ring fold
    ("k3s1_pad2_reflect", 2, 8, 10, 9, 8, 3, 1, 2, "reflect", True, None),
    ("k3s1_reflect_tiny", 1, 8, 3, 3, 8, 3, 1, 1, "reflect", False, None),
    ("k5s1_pad2_reflect", 1, 8, 12, 12, 16, 5, 1, 2, "reflect", False, None),   # 65 ring taps > 64: padded-grid path
    # the discriminator's class head: few pixels x 2 channels x K = 16 taps x 1024 -> streaming dot-product kernel
    ("k4_valid_cls_wide", 4, 1024, 4, 4, 2, 4, 1, 0, "zero", False, None),
    ("k3_thin_bias_lrelu", 2, 512, 6, 6, 5, 3, 1, 1, "zero", True, "lrelu"),
    # the 1x1 heads of the multi-scale discriminators (2048 -> 1 / 2 channels on a few pixels): streaming dot products too
    ("k1_head_2048_cls", 8, 2048, 2, 2, 2, 1, 1, 0, "zero", True, None),
    ("k1_head_2048_dis", 4, 2048, 4, 4, 1, 1, 1, 0, "zero", True, None),
    # --num_domains 3 / 5 / 6 / 7: the streaming weight-gradient kernel is instantiated for 1 / 2 / 4 / 8 rows
    ("k1_head_co3", 4, 256, 4, 4, 3, 1, 1, 0, "zero", True, None),
    ("k1_head_co5", 2, 512, 3, 3, 5, 1, 1, 0, "zero", True, None),
    ("k1_head_co7", 2, 264, 2, 2, 7, 1, 1, 0, "zero", False, None),
    # few pixels x long K (the discriminators' deep layers): forward split over the filter taps (split-K)
    ("k3s2_deep_splitk", 4, 512, 8, 8, 256, 3, 2, 1, "reflect", True, "lrelu"),
    ("k4s2_msd_splitk", 2, 256, 8, 8, 128, 4, 2, 1, "zero", False, "lrelu"),
    # ... and the data gradient of such layers (scatter form: every sub-pixel phase split over its taps; the
    # reflect-padded one writes the padded map first)
    ("k3s2_dgrad_splitk", 2, 64, 8, 8, 1024, 3, 2, 1, "reflect", True, None),
    ("k4s2_dgrad_splitk_zero", 2, 32, 8, 8, 1024, 4, 2, 1, "zero", False, "lrelu"),
    # Cout = 128 on a big map: 128 x 512 ping-pong tiles
    # (no activation: at 16.8 M outputs an fp32 ReLU mask flips on a few within-rounding-of-zero elements)
    ("k3s1_pipe512", 8, 64, 128, 128, 128, 3, 1, 1, "reflect", True, None),
    ("k3s2_pipe512", 8, 32, 256, 256, 128, 3, 2, 1, "reflect", True, None),
    ("k3s1_co64_big", 8, 64, 128, 128, 64, 3, 1, 1, "reflect", True, None),      # Cout <= 64 on a big map
    ("k3s1_co40_big", 8, 32, 128, 128, 40, 3, 1, 1, "zero", False, None),        # ... with a partial channel tile
    # many channels + bias: the two-stage bias-gradient reduction over > 512 pixel blocks
    ("k3s1_bias_big", 2, 16, 160, 160, 48, 3, 1, 1, "reflect", True, "lrelu"),
    # the straight-line epilogue (conv_device.h: epilogue_perm, round 4): absent pixels / channels are out-of-range buffer offsets,
    # the bias is range-checked -- channel counts off the tile width on the 256x256 kernels, a second channel tile that is mostly
    # padding, a ragged pixel count on the ping-pong tiles, and tanh (which keeps the general epilogue) on a wide layer
    # (no activation on the two 10^7-output cases: an fp32 mask flips on a few within-rounding-of-zero elements, see above)
    ("k3s1_tile256_co250", 16, 256, 64, 64, 250, 3, 1, 1, "reflect", True, None),
    ("k3s1_tile256_co264", 16, 64, 64, 64, 264, 3, 1, 1, "zero", True, "relu"),
    ("k3s1_pipe512_ragged", 7, 64, 100, 90, 128, 3, 1, 1, "reflect", True, None),
    ("k3s1_co100_big", 6, 40, 120, 136, 100, 3, 1, 1, "zero", True, None),
    ("k3s1_tanh_wide", 4, 64, 64, 64, 256, 3, 1, 1, "reflect", True, "tanh"),
]


def _ref_conv(x, w, b, stride, pad, pad_mode, act):
    if pad_mode == "reflect" and pad > 0:
        x = F.pad(x, (pad,) * 4, mode="reflect")
        pad = 0
    y = F.conv2d(x, w, b, stride=stride, padding=pad)
    if act == "lrelu":
        y = F.leaky_relu(y, 0.01)
    elif act == "relu":
        y = F.relu(y)
    elif act == "tanh":
        y = torch.tanh(y)
    return y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d(case, dtype, hip_device):
    ops = _ops(dtype)
    name, N, Ci, H, W, Co, k, stride, pad, pad_mode, bias, act = case
    x = _rnd(N, Ci, H, W, seed=1)
    w = _rnd(Co, Ci, k, k, seed=2, scale=(Ci * k * k) ** -0.5)
    b = _rnd(Co, seed=3, scale=0.1) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    yr = _ref_conv(xr, wr, br, stride, pad, pad_mode, act)
    gy = _rnd(*yr.shape, seed=4)
    yr.backward(gy)

    xd = x.to(hip_device).requires_grad_()
    wd = w.to(hip_device).requires_grad_()
    bd = b.to(hip_device).requires_grad_() if bias else None
    y = ops.conv2d(xd, wd, bd, stride=stride, pad=pad, pad_mode=pad_mode, act=act)
    assert ops.is_canonical(y)
    _close(y, yr, dtype, what=f"{name} fwd")
    y.backward(gy.to(hip_device))
    _close(xd.grad, xr.grad, dtype, what=f"{name} dx")
    _close(wd.grad, wr.grad, dtype, what=f"{name} dw", scale=4.0)
    if bias:
        _close(bd.grad, br.grad, dtype, what=f"{name} db", scale=4.0)


CONVT_CASES = [
    ("k3s2p1op1", 2, 16, 8, 8, 8, 3, 2, 1, 1, True, None),
    ("k3s2p1op1_odd", 1, 8, 5, 7, 16, 3, 2, 1, 1, True, None),
    ("k1s1_tanh_rgb", 2, 8, 12, 12, 3, 1, 1, 0, 0, False, "tanh"),
    ("k3s2_wide", 1, 128, 10, 10, 64, 3, 2, 1, 1, True, None),
    ("k4s2p1", 1, 8, 6, 6, 8, 4, 2, 1, 0, False, None),
    ("k1s1_tanh_rgb64", 2, 64, 24, 24, 3, 1, 1, 0, 0, False, "tanh"),     # the decoder's to-RGB layer
    ("k1s1_bias_16", 1, 16, 9, 7, 5, 1, 1, 0, 0, True, None),
    ("k3s2_pipe512", 8, 64, 64, 64, 128, 3, 2, 1, 1, True, None),     # 4 sub-pixel phases on 128 x 512 ping-pong tiles
    ("k3s2_co64_big", 8, 128, 64, 64, 64, 3, 2, 1, 1, True, None),    # the decoder's last up-conv shape class
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONVT_CASES, ids=[c[0] for c in CONVT_CASES])
def test_conv_transpose2d(case, dtype, hip_device):
    ops = _ops(dtype)
    name, N, Ci, H, W, Co, k, stride, pad, op, bias, act = case
    x = _rnd(N, Ci, H, W, seed=1)
    w = _rnd(Ci, Co, k, k, seed=2, scale=(Ci * k * k) ** -0.5)
    b = _rnd(Co, seed=3, scale=0.1) if bias else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    yr = F.conv_transpose2d(xr, wr, br, stride=stride, padding=pad, output_padding=op)
    if act == "tanh":
        yr = torch.tanh(yr)
    gy = _rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xd = x.to(hip_device).requires_grad_()
    wd = w.to(hip_device).requires_grad_()
    bd = b.to(hip_device).requires_grad_() if bias else None
    y = ops.conv_transpose2d(xd, wd, bd, stride=stride, pad=pad, out_pad=op, act=act)
    _close(y, yr, dtype, what=f"{name} fwd")
    y.backward(gy.to(hip_device))
    _close(xd.grad, xr.grad, dtype, what=f"{name} dx")
    _close(wd.grad, wr.grad, dtype, what=f"{name} dw", scale=4.0)
    if bias:
        _close(bd.grad, br.grad, dtype, what=f"{name} db", scale=4.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear(dtype, hip_device):
    ops = _ops(dtype)
    x, w, b = _rnd(6, 12, seed=1), _rnd(256, 12, seed=2), _rnd(256, seed=3)
    xr, wr, br = (t.clone().requires_grad_() for t in (x, w, b))
    yr = F.linear(xr, wr, br)
    gy = _rnd(*yr.shape, seed=4)
    yr.backward(gy)
    xd, wd, bd = (t.to(hip_device).requires_grad_() for t in (x, w, b))
    y = ops.linear(xd, wd, bd)
    _close(y, yr, torch.float32, what="linear fwd")
    y.backward(gy.to(hip_device))
    _close(xd.grad, xr.grad, torch.float32, what="linear dx")
    _close(wd.grad, wr.grad, torch.float32, what="linear dw")
    _close(bd.grad, br.grad, torch.float32, what="linear db")


def _ref_ln(x, g, b):
    shp = x.shape[1:]
    return F.layer_norm(x, shp, g.expand(shp), b.expand(shp))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["instance_relu", "instance_lrelu_res", "adain_relu", "adain_res", "layer_relu"])
@pytest.mark.parametrize("shape", [(2, 16, 12, 12), (3, 8, 5, 7), (2, 64, 33, 31)])
def test_norms(mode, shape, dtype, hip_device):
    ops = _ops(dtype)
    N, Cc, H, W = shape
    x = _rnd(*shape, seed=1) * 1.5 + 0.3
    res = _rnd(*shape, seed=5)
    gb = _rnd(N, 2 * Cc, seed=6, scale=0.5)
    gamma = (1.0 + _rnd(Cc, 1, 1, seed=7, scale=0.3))
    beta = _rnd(Cc, 1, 1, seed=8, scale=0.3)
    xr = x.clone().requires_grad_()
    resr, gbr, gr, btr = (t.clone().requires_grad_() for t in (res, gb, gamma, beta))
    xd = x.to(hip_device).requires_grad_()
    resd, gbd, gd, btd = (t.to(hip_device).requires_grad_() for t in (res, gb, gamma, beta))
    extra = []
    if mode == "instance_relu":
        yr = F.relu(F.instance_norm(xr))
        y = ops.instance_norm_act(xd, act="relu")
    elif mode == "instance_lrelu_res":
        yr = F.leaky_relu(F.instance_norm(xr), 0.01) + resr
        y = ops.instance_norm_act(xd, act="lrelu", res=resd)
        extra = [(resd, resr, "dres")]
    elif mode in ("adain_relu", "adain_res"):
        wgt, bias = torch.chunk(gbr.view(N, 2 * Cc, 1, 1), 2, dim=1)
        yr = (1 + wgt) * F.instance_norm(xr) + bias
        if mode == "adain_relu":
            yr = F.relu(yr)
            y = ops.adain_act(xd, gbd, act="relu")
        else:
            yr = yr + resr
            y = ops.adain_act(xd, gbd, res=resd)
            extra = [(resd, resr, "dres")]
        extra.append((gbd, gbr, "dgb"))
    else:
        yr = F.relu(_ref_ln(xr, gr, btr))
        y = ops.layer_norm_act(xd, gd, btd, act="relu")
        extra = [(gd, gr, "dgamma"), (btd, btr, "dbeta")]
    gy = _rnd(*shape, seed=9)
    yr.backward(gy)
    y.backward(gy.to(hip_device))
    _close(y, yr, dtype, what=f"{mode} fwd")
    _close(xd.grad, xr.grad, dtype, what=f"{mode} dx", scale=2.0)
    for a, b, nm in extra:
        sc = 8.0 if nm in ("dgb", "dgamma", "dbeta") else 1.0
        _close(a.grad, b.grad, dtype, what=f"{mode} {nm}", scale=sc)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_elementwise_and_pools(dtype, hip_device):
    ops = _ops(dtype)
    x = _rnd(2, 16, 10, 12, seed=1)
    for name, fn_ref, fn in [
        ("lrelu", lambda t: F.leaky_relu(t, 0.01), lambda t: ops.activation(t, "lrelu")),
        ("relu", F.relu, lambda t: ops.activation(t, "relu")),
        ("tanh", torch.tanh, lambda t: ops.activation(t, "tanh")),
        ("pool2", lambda t: F.avg_pool2d(t, 2, 2), ops.avg_pool2),
        ("pool3s2", lambda t: F.avg_pool2d(t, 3, 2, 1, count_include_pad=False), ops.avg_pool3s2),
        ("gap", lambda t: F.adaptive_avg_pool2d(t, 1).flatten(1), ops.global_avg_pool),
        ("add", lambda t: t + t * 0.5, lambda t: ops.add(t, ops.canon(t.detach() * 0.5))),
    ]:
        xr = x.clone().requires_grad_()
        xd = x.to(hip_device).requires_grad_()
        yr, y = fn_ref(xr), fn(xd)
        gy = _rnd(*yr.shape, seed=2)
        yr.backward(gy)
        y.backward(gy.to(hip_device))
        _close(y, yr, dtype, what=f"{name} fwd")
        ref_grad = xr.grad if name != "add" else gy  # the 0.5*t branch is detached on the device side
        _close(xd.grad, ref_grad, dtype, what=f"{name} dx")
    # odd sizes for the 3x3/s2 pool (pyramid of the multi-scale discriminator)
    x = _rnd(1, 3, 9, 7, seed=3)
    xr, xd = x.clone().requires_grad_(), x.to(hip_device).requires_grad_()
    yr, y = F.avg_pool2d(xr, 3, 2, 1, count_include_pad=False), ops.avg_pool3s2(xd)
    gy = _rnd(*yr.shape, seed=2)
    yr.backward(gy)
    y.backward(gy.to(hip_device))
    _close(y, yr, dtype, what="pool3s2 odd fwd")
    _close(xd.grad, xr.grad, dtype, what="pool3s2 odd dx")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample2_nearest(dtype, hip_device):
    """nn.Upsample(scale_factor=2, mode='nearest') (--up_type nearest), forward and gradient"""
    ops = _ops(dtype)
    for shape in ((2, 16, 6, 5), (1, 5, 3, 7)):
        x = _rnd(*shape, seed=21)
        xr, xd = x.clone().requires_grad_(), x.to(hip_device).requires_grad_()
        yr = F.interpolate(xr, scale_factor=2, mode="nearest")
        y = ops.upsample2_nearest(xd)
        g = _rnd(*yr.shape, seed=22)
        yr.backward(g)
        y.backward(g.to(hip_device))
        _close(y, yr, dtype, what="upsample2")
        _close(xd.grad, xr.grad, dtype, what="upsample2 dx", scale=4.0)


@pytest.mark.parametrize("shape", [(8, 3, 3, 3), (64, 32, 3, 3), (100, 37, 4, 4), (1024, 512, 3, 3), (2048, 1024, 4, 4)],
                         ids=["8x27", "64x288", "100x592", "1024x4608", "2048x16384"])
def test_spectral_norm_weight(shape, hip_device):
    """--dis_sn: the power iteration + W / sigma and its gradient against torch.nn.utils.spectral_norm on the CPU
    (what the reference's ConvBlock wraps its Conv2d in, functions.py:113-121): three training-mode calls (u, v carried
    in place from call to call), then an eval-mode call."""
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd.models.core.functions import conv_weight, spectral_norm
    torch.manual_seed(5)
    Co, Ci, k, _ = shape
    ref = torch.nn.utils.spectral_norm(torch.nn.Conv2d(Ci, Co, k, bias=False), "weight", 1, 1e-12, 0)
    with torch.no_grad():
        ref.weight_orig.normal_(0, 0.02)
    ours = spectral_norm(torch.nn.Conv2d(Ci, Co, k, bias=False))
    ours.load_state_dict(ref.state_dict())
    ours = ours.to(hip_device)
    x = torch.zeros(1, Ci, k, k)
    for call in range(3):
        ref.train()
        ref(x)                                  # the forward pre-hook runs the power iteration and sets ref.weight
        w_ref = ref.weight
        w = conv_weight(ours, training=True)
        g = _rnd(*shape, seed=30 + call)
        ref.weight_orig.grad = None
        w_ref.backward(g)
        ours.weight_orig.grad = None
        w.backward(g.to(hip_device))
        for name, a, b in (("u", ours.weight_u, ref.weight_u), ("v", ours.weight_v, ref.weight_v), ("weight", w, w_ref),
                           ("grad", ours.weight_orig.grad, ref.weight_orig.grad)):
            a, b = a.detach().cpu().double(), b.detach().double()
            rel = ((a - b).norm() / b.norm()).item()
            assert rel < 2e-5, f"spectral norm {shape} call {call}: {name} rel L2 {rel:.2e}"
    ref.eval()
    ref(x)
    u_before = ours.weight_u.clone()
    w = conv_weight(ours, training=False)
    assert torch.equal(u_before, ours.weight_u), "eval mode must not touch the power-iteration vectors"
    rel = ((w.detach().cpu().double() - ref.weight.detach().double()).norm() / ref.weight.detach().double().norm()).item()
    assert rel < 2e-5, f"spectral norm {shape} eval: weight rel L2 {rel:.2e}"
    # frozen discriminator (generator phases): the iteration still runs, no graph is built
    with ops.frozen(ours):
        w = conv_weight(ours, training=True)
    assert not w.requires_grad and not torch.equal(u_before, ours.weight_u)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(4, 16, 12, 12), (3, 20, 5, 7), (2, 64, 33, 31)])
def test_batch_norm(shape, dtype, hip_device):
    """nn.BatchNorm2d (--*_norm batch): training mode (batch statistics, running buffers updated in place with the
    unbiased variance), its gradients (input, weight, bias), then eval mode on the updated running statistics."""
    ops = _ops(dtype)
    N, C, H, W = shape
    ref = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        ref.weight.copy_(_rnd(C, seed=51) * 0.5 + 1.0)
        ref.bias.copy_(_rnd(C, seed=52) * 0.1)
    gw, gb = ref.weight.detach().clone().to(hip_device).requires_grad_(), ref.bias.detach().clone().to(hip_device).requires_grad_()
    rm, rv = ref.running_mean.clone().to(hip_device), ref.running_var.clone().to(hip_device)
    for call in range(2):
        x = _rnd(*shape, seed=53 + call) * 1.5 + 0.3
        xr, xd = x.clone().requires_grad_(), x.to(hip_device).requires_grad_()
        ref.train()
        ref.zero_grad()
        gw.grad = gb.grad = None
        yr = F.leaky_relu(ref(xr), 0.01)
        y = ops.batch_norm_act(xd, gw, gb, rm, rv, training=True, momentum=0.1, act="lrelu")
        g = _rnd(*shape, seed=60 + call)
        yr.backward(g)
        y.backward(g.to(hip_device))
        _close(y, yr, dtype, what="bn fwd")
        _close(xd.grad, xr.grad, dtype, what="bn dx", scale=2.0)
        _close(gw.grad, ref.weight.grad, dtype, what="bn dgamma", scale=float(N * H * W) ** 0.5)
        _close(gb.grad, ref.bias.grad, dtype, what="bn dbeta", scale=float(N * H * W) ** 0.5)
        assert (rm.cpu() - ref.running_mean).abs().max().item() < (1e-5 if dtype == torch.float32 else 5e-3)
        assert (rv.cpu() - ref.running_var).abs().max().item() < (1e-5 if dtype == torch.float32 else 1e-2)
    ref.eval()
    x = _rnd(*shape, seed=70)
    xr, xd = x.clone().requires_grad_(), x.to(hip_device).requires_grad_()
    rm0 = rm.clone()
    yr = ref(xr)
    y = ops.batch_norm_act(xd, gw, gb, rm, rv, training=False)
    yr.backward(g)
    y.backward(g.to(hip_device))
    assert torch.equal(rm0, rm), "eval mode must not touch the running statistics"
    _close(y, yr, dtype, what="bn eval fwd")
    _close(xd.grad, xr.grad, dtype, what="bn eval dx", scale=2.0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dropout(dtype, hip_device):
    """--use_dropout: x * mask / (1 - p) forward and gradient against torch with the same mask; the device-drawn mask
    is 0/1 with keep fraction 1 - p, zero in the padding channels, reproducible per (seed, offset) and different
    across seeds."""
    ops = _ops(dtype)
    x = _rnd(2, 12, 9, 7, seed=41)
    m = (torch.rand(2, 12, 9, 7, generator=torch.Generator().manual_seed(42)) < 0.5).float()
    xr, xd = x.clone().requires_grad_(), x.to(hip_device).requires_grad_()
    yr = xr * m / 0.5
    y = ops.dropout(xd, m.to(hip_device), 0.5)
    g = _rnd(*yr.shape, seed=43)
    yr.backward(g)
    y.backward(g.to(hip_device))
    _close(y, yr, dtype, what="dropout fwd", scale=2.0)
    _close(xd.grad, xr.grad, dtype, what="dropout dx", scale=2.0)
    a = ops.bernoulli_mask((4, 20, 64, 64), 0.5, 1234, 0, hip_device)
    b = ops.bernoulli_mask((4, 20, 64, 64), 0.5, 1234, 0, hip_device)
    c = ops.bernoulli_mask((4, 20, 64, 64), 0.5, 1235, 0, hip_device)
    assert ops.is_canonical(a) and a.shape == (4, 20, 64, 64)
    af = ops.to_nchw_f32(a)
    assert set(af.unique().tolist()) == {0.0, 1.0}
    assert abs(af.mean().item() - 0.5) < 5e-3                       # 327 680 draws: sigma = 9e-4
    assert torch.equal(af, ops.to_nchw_f32(b)) and not torch.equal(af, ops.to_nchw_f32(c))
    per_ch = af.mean(dim=(0, 2, 3))
    assert (per_ch - 0.5).abs().max().item() < 2e-2                 # no channel is stuck
    # padding channels (20 -> 24) stay zero
    raw = a.permute(0, 2, 3, 1)                                      # logical NHWC view of the first 20 channels
    assert raw.shape[-1] == 20
    k25 = ops.to_nchw_f32(ops.bernoulli_mask((2, 8, 32, 32), 0.25, 7, 0, hip_device)).mean().item()
    assert abs(k25 - 0.25) < 1.5e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layout_cat_noise(dtype, hip_device):
    ops = _ops(dtype)
    x = _rnd(3, 3, 8, 6, seed=1)
    xd = x.to(hip_device)
    c = ops.canon(xd)
    assert ops.is_canonical(c) and c.shape == x.shape
    _close(ops.to_nchw_f32(c), x, dtype, what="roundtrip")
    # channels_last and sliced sources
    _close(ops.to_nchw_f32(ops.canon(xd.contiguous(memory_format=torch.channels_last))), x, dtype, what="cl")
    big = _rnd(3, 7, 8, 6, seed=2).to(hip_device)
    _close(ops.to_nchw_f32(ops.canon(big[:, 2:5])), big[:, 2:5].cpu(), dtype, what="slice")
    # class planes
    cls = torch.eye(4)[[1, 3, 0]]
    xr = x.clone().requires_grad_()
    ref = torch.cat([xr, cls.view(3, 4, 1, 1).repeat(1, 1, 8, 6)], dim=1)
    xg = xd.clone().requires_grad_()
    out = ops.cat_class_planes(xg, cls.to(hip_device))
    _close(out, ref, dtype, what="cat_class")
    gy = _rnd(*ref.shape, seed=3)
    ref.backward(gy)
    out.backward(gy.to(hip_device))
    _close(xg.grad, xr.grad, dtype, what="cat_class dimg")
    # batch cat keeps layout
    a, b = ops.canon(xd[:1]), ops.canon(xd[1:])
    cb = ops.cat_batch([a, b])
    assert ops.is_canonical(cb)
    _close(cb, x, dtype, what="cat_batch")
    # device noise: N(0,1) moments, deterministic in (seed, offset)
    z = torch.zeros(4, 64, 32, 32, device=hip_device)
    n1 = ops.to_nchw_f32(ops.gaussian_noise_add(z, 1234, 0))
    n2 = ops.to_nchw_f32(ops.gaussian_noise_add(z, 1234, 0))
    n3 = ops.to_nchw_f32(ops.gaussian_noise_add(z, 1235, 0))
    assert torch.equal(n1, n2) and not torch.equal(n1, n3)
    assert abs(n1.mean().item()) < 0.01 and abs(n1.std().item() - 1.0) < 0.01
    assert abs((n1 ** 4).mean().item() - 3.0) < 0.1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_losses(dtype, hip_device):
    ops = _ops(dtype)
    # patch logits with 1 logical channel (pad channels must not count)
    x = _rnd(4, 1, 6, 6, seed=1)
    for real in (True, False):
        xr, xd = x.clone().requires_grad_(), x.to(hip_device).requires_grad_()
        t = torch.ones_like(x) if real else torch.zeros_like(x)
        lr = F.binary_cross_entropy_with_logits(xr, t)
        l = ops.bce_logits_const(xd, real)
        (lr * 3).backward()
        (l * 3).backward()
        _close(l, lr, dtype, what="bce_const")
        _close(xd.grad, xr.grad, dtype, what="bce_const dx")
    # the other GANLoss modes: lsgan (MSE vs ones / zeros) and the hinge terms written out in the reference model
    for what, ref_fn, dev_fn in (
            ("lsgan_real", lambda v: F.mse_loss(v, torch.ones_like(v)), lambda v: ops.mse_const(v, True)),
            ("lsgan_fake", lambda v: F.mse_loss(v, torch.zeros_like(v)), lambda v: ops.mse_const(v, False)),
            ("hinge_real", lambda v: F.relu(1.0 - v).mean(), lambda v: ops.hinge_dis(v, True)),
            ("hinge_fake", lambda v: F.relu(1.0 + v).mean(), lambda v: ops.hinge_dis(v, False)),
            ("hinge_gen", lambda v: -v.mean(), lambda v: ops.neg_mean(v))):
        xr, xd = x.clone().requires_grad_(), x.to(hip_device).requires_grad_()
        lr, l = ref_fn(xr), dev_fn(xd)
        (lr * 3).backward()
        (l * 3).backward()
        _close(l, lr, dtype, what=what)
        _close(xd.grad, xr.grad, dtype, what=what + " dx")
    # relativistic-average term: x - mean(other), gradient to both
    o = _rnd(4, 1, 6, 6, seed=11)
    xr, orr = x.clone().requires_grad_(), o.clone().requires_grad_()
    xd, od = x.to(hip_device).requires_grad_(), o.to(hip_device).requires_grad_()
    lr = F.binary_cross_entropy_with_logits(xr - torch.mean(orr), torch.ones_like(xr))
    l = ops.bce_logits_const(ops.sub_mean(xd, od), True)
    lr.backward()
    l.backward()
    _close(l, lr, dtype, what="sub_mean loss")
    _close(xd.grad, xr.grad, dtype, what="sub_mean dx")
    _close(od.grad, orr.grad, dtype, what="sub_mean dother")
    p, t = _rnd(6, 4, seed=2), torch.eye(4)[[0, 1, 2, 3, 1, 2]]
    pr, pd = p.clone().requires_grad_(), p.to(hip_device).requires_grad_()
    lr, l = F.binary_cross_entropy_with_logits(pr, t), ops.bce_logits(pd, t.to(hip_device))
    lr.backward()
    l.backward()
    _close(l, lr, torch.float32, what="bce_target")
    _close(pd.grad, pr.grad, torch.float32, what="bce_target dx")
    a, b = _rnd(2, 3, 16, 16, seed=3), _rnd(2, 3, 16, 16, seed=4)
    ar, ad = a.clone().requires_grad_(), a.to(hip_device).requires_grad_()
    lr, l = F.l1_loss(b, ar) * 10, ops.l1_loss(b.to(hip_device), ad) * 10
    lr.backward()
    l.backward()
    _close(l, lr, dtype, what="l1")
    _close(ad.grad, ar.grad, dtype, what="l1 dx")
    z = _rnd(2, 16, 8, 8, seed=5)
    zr, zd = z.clone().requires_grad_(), z.to(hip_device).requires_grad_()
    lr, l = torch.mean(torch.pow(zr, 2)) * 0.01, ops.l2_mean(zd) * 0.01
    lr.backward()
    l.backward()
    _close(l, lr, dtype, what="l2mean")
    _close(zd.grad, zr.grad, dtype, what="l2mean dx")
    mu, lv, eps = _rnd(4, 8, seed=6), _rnd(4, 8, seed=7) * 0.3, _rnd(4, 8, seed=8)
    mr, lvr = mu.clone().requires_grad_(), lv.clone().requires_grad_()
    md, lvd = mu.to(hip_device).requires_grad_(), lv.to(hip_device).requires_grad_()
    zr = eps * torch.exp(0.5 * lvr) + mr
    klr = torch.sum(1 + lvr - mr.pow(2) - lvr.exp()) * -0.5 * 0.01
    z_ = ops.reparameterize(md, lvd, eps.to(hip_device))
    kl = ops.kl_sum(md, lvd) * 0.01
    g = _rnd(4, 8, seed=9)
    (zr * g).sum().add(klr).backward()
    ((z_ * g.to(hip_device)).sum() + kl).backward()
    _close(z_, zr, torch.float32, what="reparam")
    _close(kl, klr, torch.float32, what="kl")
    _close(md.grad, mr.grad, torch.float32, what="dmu")
    _close(lvd.grad, lvr.grad, torch.float32, what="dlogvar")
    l1r = F.l1_loss(mr[:2], eps[:2])
    l1 = ops.l1_loss(md[:2], eps[:2].to(hip_device))
    _close(l1, l1r, torch.float32, what="l1 2d")


def test_adam_multi(hip_device):
    from masterthesis_amd import hip_ops as ops
    torch.manual_seed(0)
    ps = [torch.randn(s) for s in [(16, 3, 3, 3), (7,), (1000, 33)]]
    ref = [p.clone().requires_grad_() for p in ps]
    opt = torch.optim.Adam(ref, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    dev = [p.to(hip_device) for p in ps]
    ms = [torch.zeros_like(p) for p in dev]
    vs = [torch.zeros_like(p) for p in dev]
    for step in range(1, 4):
        gs = [torch.randn_like(p) for p in ps]
        for r, g in zip(ref, gs):
            r.grad = g.clone()
        opt.step()
        ops.adam_multi(dev, [g.to(hip_device) for g in gs], ms, vs, 1e-3, 0.5, 0.999, 1e-8, 1e-4, step)
    for d, r in zip(dev, ref):
        _close(d, r, torch.float32, what="adam")


def test_pack_multi_matches_single(hip_device):
    """mt_conv_pack_multi_* (one launch for many weight images) writes the same bytes as mt_conv_pack."""
    import ctypes as C
    from masterthesis_amd import _lib as L
    lib = L.load()
    cases = [  # dtype, transposed, Ci, Co, k, stride, pad, pad_mode, which
        (L.MT_BF16, 0, 16, 24, 3, 1, 1, L.PAD_REFLECT, L.PACK_FWD),
        (L.MT_BF16, 0, 16, 24, 3, 1, 1, L.PAD_REFLECT, L.PACK_BWD_DATA),     # ring image (21 taps)
        (L.MT_BF16, 0, 8, 40, 3, 2, 1, L.PAD_REFLECT, L.PACK_BWD_DATA),      # 4 sub-pixel phase images
        (L.MT_F32, 1, 24, 8, 3, 2, 1, L.PAD_ZERO, L.PACK_FWD),               # transposed conv, phased
        (L.MT_F32, 0, 3, 8, 7, 1, 3, L.PAD_REFLECT, L.PACK_BWD_DATA),
        # big bf16 tensors with k <= 4 take the grouped LDS-tiled kernel (fp32 images and 7x7 stay element-wise)
        (L.MT_BF16, 0, 128, 96, 3, 1, 1, L.PAD_REFLECT, L.PACK_FWD),          # mode 1 (taps contiguous per row)
        (L.MT_BF16, 0, 128, 96, 3, 1, 1, L.PAD_REFLECT, L.PACK_BWD_DATA),     # mode 2, ring image with repeated taps
        (L.MT_BF16, 0, 100, 120, 3, 1, 1, L.PAD_REFLECT, L.PACK_FWD),         # ragged: 100 -> 104 padded columns
        (L.MT_F32, 0, 100, 120, 3, 1, 1, L.PAD_ZERO, L.PACK_BWD_DATA),        # ragged rows, fp32 image
        (L.MT_BF16, 0, 128, 256, 3, 2, 1, L.PAD_REFLECT, L.PACK_BWD_DATA),    # 4 sub-pixel phase images (1-4 taps)
        (L.MT_BF16, 1, 256, 128, 3, 2, 1, L.PAD_ZERO, L.PACK_FWD),            # IOHW weights, phased forward images
        (L.MT_BF16, 0, 64, 128, 4, 2, 1, L.PAD_ZERO, L.PACK_FWD),             # 16 taps
        (L.MT_BF16, 0, 64, 128, 4, 2, 1, L.PAD_ZERO, L.PACK_BWD_DATA),
        (L.MT_BF16, 0, 32, 64, 7, 1, 3, L.PAD_REFLECT, L.PACK_FWD),           # 49 taps
    ]
    descs, ws, singles, multis, whichs = [], [], [], [], []
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for i, (dt, tr, ci, co, k, st, pad, pm, which) in enumerate(cases):
        d = L.ConvDesc(dt, tr, 2, 16, 16, ci, co, k, k, st, pad, pm, 1 if tr else 0, L.ACT_NONE, 0.0)
        shape = (ci, co, k, k) if tr else (co, ci, k, k)
        w = torch.randn(*shape, generator=torch.Generator().manual_seed(i)).to(hip_device)
        n = int(lib.mt_conv_pack_bytes(C.byref(d), which))
        a = torch.zeros(n, dtype=torch.uint8, device=hip_device)
        b = torch.zeros(n, dtype=torch.uint8, device=hip_device)
        L.check(lib.mt_conv_pack(C.byref(d), which, C.c_void_p(w.data_ptr()), C.c_void_p(a.data_ptr()), stream), "pack")
        descs.append(d); ws.append(w); singles.append(a); multis.append(b); whichs.append(which)
    n = len(cases)
    host = C.create_string_buffer(int(lib.mt_conv_pack_multi_table_bytes(n)))
    ne, nb = C.c_int(), C.c_int()
    L.check(lib.mt_conv_pack_multi_build(n, (L.ConvDesc * n)(*descs), (C.c_int * n)(*whichs),
                                         (C.c_void_p * n)(*[w.data_ptr() for w in ws]),
                                         (C.c_void_p * n)(*[b.data_ptr() for b in multis]), host, C.byref(ne),
                                         C.byref(nb)), "build")
    # (the two counts are opaque to the caller since round 3: element-wise entries | grouped tensors << 16 -- the big bf16
    #  k <= 4 tensors go through the grouped LDS-tiled kernel, one group per weight tensor)
    assert (ne.value & 0xffff) >= 11 and (ne.value >> 16) == 7
    dev = torch.frombuffer(host, dtype=torch.uint8).clone().to(hip_device)
    L.check(lib.mt_conv_pack_multi_run(C.c_void_p(dev.data_ptr()), ne.value, nb.value, stream), "run")
    torch.cuda.synchronize()
    for a, b in zip(singles, multis):
        assert torch.equal(a.cpu(), b.cpu())


def test_zero_arena_scope(hip_device):
    """transient zero buffers: zero at hand-out inside a scope, re-zeroed by the next scope, plain zeros outside."""
    from masterthesis_amd import hip_ops as ops
    arena = ops._ZeroArena()
    t = arena.take((4, 8), hip_device)                      # outside a scope: ordinary allocation
    assert t.sum().item() == 0
    arena.begin(hip_device)
    a = arena.take((3, 5), hip_device)
    assert a.abs().sum().item() == 0
    a.fill_(7.0)
    arena.end()
    arena.begin(hip_device)
    b = arena.take((3, 5), hip_device)
    assert b.data_ptr() == a.data_ptr() and b.abs().sum().item() == 0      # same memory, zero again
    big = arena.take((1 << 22,), hip_device)                # larger than the arena: falls back, still zero
    assert big.abs().sum().item() == 0
    arena.end()
    arena.begin(hip_device)                                 # the arena grew to the size that was asked for
    c = arena.take((1 << 22,), hip_device)
    assert c.abs().sum().item() == 0 and arena.off >= (1 << 22)
    arena.end()
