"""One-pass backward of InstanceNorm / AdaIN (mt_norm_bwd_onepass: statistics + coefficients + apply in one launch, the plane of
(image, 16 channels) resident in registers) against (a) the three-launch backward on the same device tensors -- same arithmetic,
different fp32 summation order -- and (b) the fp32 PyTorch CPU reference of the op (reference functions.py:17, norm.py:29-33)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# name, N, C, H, W, mode, act, residual            (one-pass needs N * H * W * C / 65536 >= 128 workgroups)
CASES = [
    ("in_relu_k1", 8, 256, 64, 64, "instance", "relu", False),          # the K1 site: 16 slices per image, 32 chunks per pixel
    ("adain_res_k1", 8, 256, 64, 64, "adain", None, True),              # decoder residual block: AdaIN + skip
    ("adain_relu_32x32", 32, 256, 32, 32, "adain", "relu", False),      # 4 slices per image
    ("in_lrelu_512ch", 8, 512, 32, 64, "instance", "lrelu", True),      # 64 chunks per pixel, non-square
    ("in_relu_64ch_256", 2, 64, 256, 256, "instance", "relu", False),   # 8 chunks per pixel, 64 slices per image
    ("adain_128ch_128", 4, 128, 128, 128, "adain", "relu", False),      # 16 chunks per pixel, 32 slices per image
    ("adain_relu_c248", 16, 248, 32, 32, "adain", "relu", False),       # 31 chunks per pixel: not a power of two -> three-pass
    ("in_relu_small", 2, 256, 64, 64, "instance", "relu", False),       # 32 workgroups only -> three-pass
]


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()


def _run(ops, name, N, C, H, W, mode, act, res, dev, onepass):
    ops.set_norm_onepass(onepass)
    try:
        x = (_rnd(N, C, H, W, seed=1) * 1.5 + 0.3).to(dev).requires_grad_()
        r = _rnd(N, C, H, W, seed=2).to(dev).requires_grad_() if res else None
        gb = _rnd(N, 2 * C, seed=3, scale=0.5).to(dev).requires_grad_()
        ops.hbm_timer_start()
        if mode == "instance":
            y = ops.instance_norm_act(x, act=act, res=r)
        else:
            y = ops.adain_act(x, gb, act=act, res=r)
        y.backward(_rnd(N, C, H, W, seed=4).to(dev))
        torch.cuda.synchronize()
        used = ops.hbm_timer_stop()
        return y, x.grad, (gb.grad if mode == "adain" else None), (r.grad if res else None), used
    finally:
        ops.set_norm_onepass(True)


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_onepass_matches_three_pass_and_reference(case, hip_device):
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    name, N, C, H, W, mode, act, res = case
    y1, dx1, dgb1, dr1, used1 = _run(ops, *case, hip_device, True)
    y3, dx3, dgb3, dr3, used3 = _run(ops, *case, hip_device, False)
    from masterthesis_amd import _lib as L
    expect = bool(L.load().mt_norm_bwd_onepass_ok(L.MT_BF16, L.NORM_INSTANCE if mode == "instance" else L.NORM_ADAIN, N, H * W,
                                                  ops.padc(C), {None: L.ACT_NONE, "relu": L.ACT_RELU, "lrelu": L.ACT_LRELU}[act],
                                                  0, None))
    assert expect == (name not in ("adain_relu_c248", "in_relu_small"))
    assert ("norm_bwd_onepass" in used1) == expect, used1.keys()
    assert "norm_bwd_onepass" not in used3 and "norm_bwd_apply" in used3
    assert torch.equal(y1, y3)
    # same formula, sums in a different order: dx differs by bf16 output rounding of a few elements at most
    d = (dx1.float() - dx3.float()).abs()
    ref = dx3.float().abs().max().item()
    assert d.max().item() <= 2 ** -7 * ref + 1e-6, (name, d.max().item(), ref)
    assert (d > 0).float().mean().item() < 0.02, (name, (d > 0).float().mean().item())
    if dgb1 is not None:
        assert torch.allclose(dgb1, dgb3, rtol=2e-4, atol=2e-3 * dgb3.abs().max().item()), (dgb1 - dgb3).abs().max()
    if dr1 is not None:
        assert torch.equal(dr1, dr3)
    # fp32 CPU reference of the op
    xr = (_rnd(N, C, H, W, seed=1) * 1.5 + 0.3).requires_grad_()
    gbr = _rnd(N, 2 * C, seed=3, scale=0.5).requires_grad_()
    yr = F.instance_norm(xr)
    if mode == "adain":
        wgt, bias = torch.chunk(gbr.view(N, 2 * C, 1, 1), 2, dim=1)
        yr = (1 + wgt) * yr + bias
    if act == "relu":
        yr = F.relu(yr)
    elif act == "lrelu":
        yr = F.leaky_relu(yr, 0.01)
    yr.backward(_rnd(N, C, H, W, seed=4))
    num = (dx1.float().cpu() - xr.grad).norm().item()
    den = xr.grad.norm().item()
    assert num <= 4e-2 * den, (name, num / den)          # (bound of tests/test_ops_gpu.py::test_norms for bf16 dx)
    if mode == "adain":
        num = (dgb1.cpu() - gbr.grad).norm().item()
        assert num <= 8e-2 * gbr.grad.norm().item(), (name, num / gbr.grad.norm().item())   # (test_norms: scale 8 for dgb)


def test_slices_per_image_are_bounded_by_residency(hip_device):
    """every slice of an image waits for the others, so an image may have at most as many slices as workgroups
    of the kernel can be resident (occupancy x compute units) -- minus what the caller reserves for an overlapping collective."""
    import ctypes as C
    from masterthesis_amd import hip_ops as ops, _lib as L
    lib = L.load()
    cap = int(lib.mt_norm_bwd_onepass_capacity())
    props = torch.cuda.get_device_properties(hip_device)
    assert cap >= props.multi_processor_count and cap % props.multi_processor_count == 0, cap
    sl = C.c_int(0)
    # 64 channels (8 chunks per pixel): slices per image = HW / 1024
    ok = lambda hw, lim: bool(lib.mt_norm_bwd_onepass_ok(L.MT_BF16, L.NORM_INSTANCE, 2, hw, 64, L.ACT_RELU, lim, C.byref(sl)))
    assert ok(cap * 1024, 0) and sl.value == cap                        # S = capacity: taken
    assert not ok((cap + 1) * 1024, 0) and sl.value == 0                # S = capacity + 1: refused
    assert not ok(cap * 1024, cap - 1)                                  # the caller's limit counts
    assert ok((cap - 64) * 1024, cap - 64) and not ok((cap - 63) * 1024, cap - 64)
    assert not ok(cap * 1024, 10 * cap) or sl.value == cap              # a limit above the capacity does not raise it
    assert not ok((cap + 1) * 1024, 10 * cap)
    # the op layer honours the reserve: a 512 x 512 plane of 64 channels (256 slices) falls back to three launches
    ops.set_compute_dtype(torch.bfloat16)
    try:
        ops.set_onepass_reserve(64)
        y, dx, _, _, used = _run(ops, "reserve", 1, 64, 512, 512, "instance", "relu", False, hip_device, True)
        assert "norm_bwd_onepass" not in used and "norm_bwd_apply" in used
        ops.set_onepass_reserve(0)
        y2, dx2, _, _, used2 = _run(ops, "reserve", 1, 64, 512, 512, "instance", "relu", False, hip_device, True)
        assert ("norm_bwd_onepass" in used2) == (cap >= 256)
        assert (dx.float() - dx2.float()).abs().max().item() <= 2 ** -7 * dx.float().abs().max().item() + 1e-6
    finally:
        ops.set_onepass_reserve(0)
    ops.check_device_status(hip_device)


def test_give_up_reaches_the_host(hip_device):
    """a wait that gives up must not be a silent NaN.  The arrive counter of one image is preset so that it
    can never reach the slice count (2^32 - S: the last arrival wraps it to 0), the poll bound is a kernel argument (64 here), and
    the status word the kernel sets surfaces as a RuntimeError -- through check_device_status and through Model.sync_losses'
    single device->host copy.  The kernel still terminates and leaves the counters clean."""
    import ctypes as C
    from masterthesis_amd import hip_ops as ops, _lib as L
    lib = L.load()
    N, Cc, H, W = 8, 256, 64, 64
    S = (H * W * Cc // 8) // 8192
    P = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    x = ops.canon(_rnd(N, Cc, H, W, seed=1).to(hip_device), torch.bfloat16)
    dy = ops.canon(_rnd(N, Cc, H, W, seed=2).to(hip_device), torch.bfloat16)
    coef = torch.ones(4, N, Cc, device=hip_device)
    dx = torch.empty_like(x)
    part = torch.empty(N, S, Cc, 2, device=hip_device)
    sync = torch.zeros(2 * N, dtype=torch.int32, device=hip_device)
    sync[3] = -S                                    # image 3 can never complete
    status = torch.zeros(4, dtype=torch.int32, device=hip_device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.mt_norm_bwd_onepass(L.MT_BF16, L.NORM_INSTANCE, P(dy), P(x), P(coef[0]), P(coef[1]), P(coef[2]), P(coef[3]), None,
                                    None, None, None, None, P(dx), P(part), P(sync), P(status), 64, N, H * W, Cc, Cc, L.ACT_RELU, 0.0, st), "onepass")
    torch.cuda.synchronize()
    words = status.cpu().tolist()
    assert words[0] & 1 and 3 * S < words[1] <= 4 * S, words          # a workgroup of image 3 reported
    assert int((sync != 0).sum()) == 0                                  # self-cleaning also after a give-up
    nan_img = torch.isnan(dx.float()).flatten(1).any(1).cpu().tolist()
    assert nan_img[3] and not any(nan_img[:3]) and not any(nan_img[4:]), nan_img
    with pytest.raises(RuntimeError, match="one-pass norm backward"):
        ops.raise_on_device_status(words)
    # through the op layer: the same failure injected into the module's own status words reaches sync_losses
    dev_words = ops.device_status(hip_device)
    dev_words[:2] = torch.tensor([1, 7], dtype=torch.int32, device=hip_device)
    with pytest.raises(RuntimeError, match="workgroup 6 gave up"):
        ops.check_device_status(hip_device)
    assert int(dev_words.abs().sum()) == 0                              # cleared for the next step
    ops.check_device_status(hip_device)


def test_launches_from_two_streams_are_ordered(hip_device):
    """two one-pass launches must never be resident together.  The op layer orders a launch from another
    stream behind the stream of the previous launch; results on both streams equal the single-stream ones."""
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    case = ("in_relu_k1", 8, 256, 64, 64, "instance", "relu", False)
    _, ref, _, _, used = _run(ops, *case, hip_device, True)
    assert "norm_bwd_onepass" in used
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(3):
        for st in (s1, s2):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                outs.append(_run(ops, *case, hip_device, True)[1])
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)
    ops.check_device_status(hip_device)


LN_CASES = [
    ("ln_relu_128ch_128", 4, 128, 128, 128, "relu"),        # dec2.0 (reference networks.py:248-249): 32 slices per image
    ("ln_relu_64ch_256", 2, 64, 256, 256, "relu"),          # dec2.1: 64 slices per image
    ("ln_none_256ch_64", 8, 256, 64, 64, None),
]


@pytest.mark.parametrize("case", LN_CASES, ids=[c[0] for c in LN_CASES])
def test_onepass_layer_norm_matches_three_pass_and_reference(case, hip_device):
    """Round 4: the reference's per-sample LayerNorm (norm.py:5-21; its backward was the reference's CPU outlier) on the one-pass
    kernel: dx, dgamma, dbeta against the three-launch backward on the same device tensors and against the fp32 CPU reference."""
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    name, N, C, H, W, act = case
    x0 = (_rnd(N, C, H, W, seed=1) * 1.5 + 0.3)
    g0 = _rnd(C, 1, 1, seed=2, scale=0.5) + 1.0
    b0 = _rnd(C, 1, 1, seed=3, scale=0.2)
    gy = _rnd(N, C, H, W, seed=4)
    outs = {}
    for onepass in (True, False):
        ops.set_norm_onepass(onepass)
        try:
            x = x0.to(hip_device).requires_grad_()
            gm = g0.to(hip_device).requires_grad_()
            bt = b0.to(hip_device).requires_grad_()
            ops.hbm_timer_start()
            y = ops.layer_norm_act(x, gm, bt, act=act)
            y.backward(gy.to(hip_device))
            torch.cuda.synchronize()
            used = ops.hbm_timer_stop()
            outs[onepass] = (y.detach(), x.grad.float().cpu(), gm.grad.cpu(), bt.grad.cpu(), used)
        finally:
            ops.set_norm_onepass(True)
    y1, dx1, dg1, db1, used1 = outs[True]
    y3, dx3, dg3, db3, used3 = outs[False]
    assert "norm_bwd_onepass" in used1 and "norm_bwd_onepass" not in used3 and "norm_bwd_apply" in used3
    assert torch.equal(y1, y3)
    d = (dx1 - dx3).abs()
    ref = dx3.abs().max().item()
    assert d.max().item() <= 2 ** -7 * ref + 1e-6 and (d > 0).float().mean().item() < 0.02, (name, d.max().item(), ref)
    assert torch.allclose(dg1, dg3, rtol=2e-4, atol=2e-4 * dg3.abs().max().item())
    assert torch.allclose(db1, db3, rtol=2e-4, atol=2e-4 * db3.abs().max().item())
    xr, gr, br = x0.clone().requires_grad_(), g0.clone().requires_grad_(), b0.clone().requires_grad_()
    shp = xr.shape[1:]
    yr = F.layer_norm(xr, shp, gr.expand(shp), br.expand(shp))
    if act == "relu":
        yr = F.relu(yr)
    yr.backward(gy)
    assert ((dx1 - xr.grad).norm() / xr.grad.norm()).item() <= 4e-2
    # (bf16 x / dy / dx storage: 2.1e-2 measured on dbeta of the 64-channel case, the three-launch backward gives the same value)
    assert ((dg1 - gr.grad).norm() / gr.grad.norm()).item() <= 4e-2
    assert ((db1 - br.grad).norm() / br.grad.norm()).item() <= 4e-2
    ops.check_device_status(hip_device)
