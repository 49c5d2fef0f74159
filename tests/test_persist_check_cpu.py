"""The comparison used by tests/test_persist_gpu.py, validated without a GPU.

A stand-in for the device path is built on the CPU from the same arithmetic contract -- bf16 operands, fp32
accumulation in ANOTHER summation order (input channels permuted), bf16 storage of y, of dz = gy * act'(y) and of
dx -- and pushed through `check_against_reference` for every case of CASES that has a fused activation, over several
seeds.  The activation masks of the stand-in and of the reference do differ in a few places (asserted: otherwise this
test would not be testing anything), which is exactly what broke the element-wise bound of round 2; the bound through
the device's own mask must hold regardless, and a misplaced tile must still be caught."""
import pytest
import torch
import torch.nn.functional as F

from test_persist_gpu import CASES, check_against_reference, reference, case_seed

ACT_CASES = [c for c in CASES if c[12] is not None]


def _shrunk(case, n=2, hw=64):
    name, kind, N, Ci, H, W, Co, k, stride, pad, pad_mode, bias, act = case
    return (name, kind, min(N, n), Ci, min(H, hw), min(W, hw + 1), Co, k, stride, pad, pad_mode, bias, act)


def _standin(case, seed, flip=0):
    name, kind, N, Ci, H, W, Co, k, stride, pad, pad_mode, bias, act = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Co, Ci, k, k, generator=g) * (Ci * k * k) ** -0.5).bfloat16().float()
    b = (torch.randn(Co, generator=g) * 0.1) if bias else None
    perm = torch.randperm(Ci, generator=g)
    xr = x[:, perm].clone().requires_grad_()
    xp = F.pad(xr, (pad,) * 4, mode="reflect") if (pad_mode == "reflect" and pad) else xr
    # another summation order: channels permuted and the reduction split in two halves added afterwards
    wp = w[:, perm]
    h = max(1, Ci // 2)
    p0 = 0 if (pad_mode == "reflect" and pad) else pad
    pre = F.conv2d(xp[:, :h], wp[:, :h], None, stride=stride, padding=p0)
    if h < Ci:
        pre = pre + F.conv2d(xp[:, h:], wp[:, h:], None, stride=stride, padding=p0)
    if b is not None:
        pre = pre + b.view(1, -1, 1, 1)
    slope = {"relu": 0.0, "lrelu": 0.01}[act]
    pre_d = pre.detach().clone()
    if flip:
        # what a rounding-level difference does at full size (25 M outputs: 1-3 per run): the pre-activations closest
        # to zero come out with the other sign
        idx = pre_d.abs().flatten().topk(flip, largest=False).indices
        pre_d.view(-1)[idx] = -pre_d.view(-1)[idx]
    y_dev = torch.where(pre_d > 0, pre_d, pre_d * slope).bfloat16().float()
    gy = torch.randn(*y_dev.shape, generator=g).bfloat16().float()
    dz = (gy * torch.where(y_dev > 0, torch.ones_like(gy), torch.full_like(gy, slope))).bfloat16().float()
    (dxp,) = torch.autograd.grad(pre, xr, dz)
    dx_dev = torch.empty_like(dxp)
    dx_dev[:, perm] = dxp
    return x, w, b, gy, y_dev, dx_dev.bfloat16().float()


@pytest.mark.parametrize("case", ACT_CASES, ids=[c[0] for c in ACT_CASES])
def test_bound_holds_for_another_summation_order(case):
    case = _shrunk(case)
    old_bound_broken = 0
    for seed in [case_seed(case[0])] + list(range(5)):
        x, w, b, gy, y_dev, dx_dev = _standin(case, seed, flip=3)
        check_against_reference(case, x, w, b, gy, y_dev, dx_dev)
        yr, dx_ref, _, pre = reference(case, x, w, b, gy, y_dev)
        assert int(((pre > 0) != (y_dev > 0)).sum()) >= 1
        old_bound_broken += (dx_dev - dx_ref).abs().max().item() > 0.03 * dx_ref.abs().max().item() + 0.02
    if case[12] == "relu":
        # round 2's bound (through the REFERENCE's mask) does break on such inputs: that was the red test
        assert old_bound_broken > 0


def test_misplaced_tile_is_still_caught():
    case = _shrunk(ACT_CASES[0])
    x, w, b, gy, y_dev, dx_dev = _standin(case, 0)
    bad = dx_dev.clone()
    bad[0, :, 8:16, 8:24] = dx_dev[0, :, 24:32, 8:24]           # a 128-pixel tile's worth of rows from elsewhere
    with pytest.raises(AssertionError, match="dx"):
        check_against_reference(case, x, w, b, gy, y_dev, bad)
    bad = y_dev.clone()
    bad[1, :, 0:2] = 0
    with pytest.raises(AssertionError, match="fwd"):
        check_against_reference(case, x, w, b, gy, bad, dx_dev)
