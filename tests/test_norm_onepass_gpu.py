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
                                                  None))
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
