"""Grouped weight gradient (mt_conv_bwd_weight_group / hip_ops' deferred weight-gradient queue): up to four convolutions of the
same descriptor share one launch of the 256x256 weight-gradient kernel (28 / G pixel splits each instead of 28).  Checked against
the one-problem path on the same device tensors (same products, fp32 sums in a different order) and against the fp32 CPU
reference of nn.Conv2d's weight gradient (reference blocks.py:131-132,150-151)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).bfloat16().float()


# name, N, C, H, W, pad_mode, G       (k1_*: splits longer than 6976 pixels run the 16-bit / 8-bit pixel-delta tables)
CASES = [
    ("small_reflect_g4", 4, 256, 32, 32, "reflect", 4),
    ("small_zero_g2", 8, 256, 32, 32, "zero", 2),
    ("small_reflect_g3", 8, 256, 32, 32, "reflect", 3),
    ("k1_reflect_g4", 16, 256, 64, 64, "reflect", 4),
    ("k1_zero_g3", 16, 256, 64, 64, "zero", 3),
    ("k1_reflect_g7", 16, 256, 64, 64, "reflect", 7),       # 4 splits of 16384 pixels: the 8-bit delta table
    ("k1n32_zero_g4", 32, 256, 64, 64, "zero", 4),          # 7 splits of 18725 pixels: 8-bit deltas, zero padding
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_group_matches_single_and_reference(case, hip_device):
    from masterthesis_amd import hip_ops as ops, _lib as L
    ops.set_compute_dtype(torch.bfloat16)
    lib = L.load()
    name, N, Cc, H, W, pad_mode, G = case
    desc = L.ConvDesc(L.MT_BF16, 0, N, H, W, Cc, Cc, 3, 3, 1, 1, L.PAD_REFLECT if pad_mode == "reflect" else L.PAD_ZERO, 0,
                      L.ACT_NONE, 0.0)
    assert lib.mt_conv_bwd_weight_group_max(C.byref(desc)) >= G
    xs = [ops.canon(_rnd(N, Cc, H, W, seed=10 + g).to(hip_device)) for g in range(G)]
    dys = [ops.canon(_rnd(N, Cc, H, W, seed=20 + g, scale=0.5).to(hip_device)) for g in range(G)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    # one by one
    single = []
    nws = int(lib.mt_conv_bwd_weight_ws_bytes(C.byref(desc)))
    ws = torch.empty((nws,), dtype=torch.uint8, device=hip_device)
    for g in range(G):
        dw = torch.empty(Cc, Cc, 3, 3, dtype=torch.float32, device=hip_device)
        L.check(lib.mt_conv_bwd_weight(C.byref(desc), P(xs[g]), P(dys[g]), P(dw), None, P(ws), nws, 0, st), "single")
        single.append(dw)
    # grouped, accumulate = 1 on top of a known start
    nwg = int(lib.mt_conv_bwd_weight_group_ws_bytes(C.byref(desc), G))
    assert nwg > 0
    wsg = torch.empty((nwg,), dtype=torch.uint8, device=hip_device)
    start = [torch.full((Cc, Cc, 3, 3), 0.25 * (g + 1), dtype=torch.float32, device=hip_device) for g in range(G)]
    grouped = [s.clone() for s in start]
    xa = (C.c_void_p * G)(*[t.data_ptr() for t in xs])
    da = (C.c_void_p * G)(*[t.data_ptr() for t in dys])
    wa = (C.c_void_p * G)(*[t.data_ptr() for t in grouped])
    L.check(lib.mt_conv_bwd_weight_group(C.byref(desc), G, xa, da, wa, P(wsg), nwg, 1, st), "group")
    torch.cuda.synchronize()
    for g in range(G):
        got = grouped[g] - start[g]
        ref = single[g]
        err = (got - ref).abs().max().item()
        assert err <= 2e-5 * ref.abs().max().item() + 1e-4, (name, g, err, ref.abs().max().item())
    # fp32 CPU reference of the first and the last problem (the operands are bf16-exact, products are exact in fp32)
    for g in ((0, G - 1) if N <= 16 else (G - 1,)):
        x = ops.to_nchw_f32(xs[g]).cpu()
        dy = ops.to_nchw_f32(dys[g]).cpu()
        xp = F.pad(x, (1, 1, 1, 1), mode="reflect") if pad_mode == "reflect" else x
        w = torch.zeros(Cc, Cc, 3, 3, requires_grad=True)
        F.conv2d(xp, w, None, 1, 0 if pad_mode == "reflect" else 1).backward(dy)
        got = (grouped[g] - start[g]).cpu()
        rel = ((got - w.grad).norm() / w.grad.norm()).item()
        assert rel < 2e-4, (name, g, rel)


def test_deferred_queue_groups_inside_backward(hip_device):
    """Six convolutions of one geometry in a chain: with the queue on, backward launches one group of four and (from the engine
    callback) one of two; parameter gradients equal the ungrouped run; the ready hook of every weight fires exactly once."""
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(3)
    ws = [(torch.randn(256, 256, 3, 3) * 0.02).to(hip_device).requires_grad_() for _ in range(6)]
    x0 = _rnd(8, 256, 32, 32, seed=5).to(hip_device)
    gy = ops.canon(_rnd(8, 256, 32, 32, seed=6).to(hip_device))
    res = {}
    fused_before = ops._FUSE_WGRAD_ACC[0]
    ops.set_fused_grad_accumulation(True)           # gradients go straight into param.grad (what FusedAdam sets up)
    try:
        for on in (False, True):
            ops.set_wgrad_group(on)
            fired = []
            for w in ws:
                w.grad = torch.zeros_like(w)              # fused accumulation target
                ops.set_grad_ready_hook(w, lambda p, fired=fired: fired.append(id(p)))
            x = x0.clone().requires_grad_()
            h = x
            for w in ws:
                h = ops.conv2d(h, w, None, stride=1, pad=1, pad_mode="reflect", act="relu")
            ops.oplog_start()
            h.backward(gy)
            ev = ops.oplog_stop()
            groups = sorted(d[16] for k, d, ms in ev if k == "wgrad" and len(d) > 16)
            res[on] = ([w.grad.clone() for w in ws], ops.to_nchw_f32(x.grad), groups, sorted(fired))
        # six problems: full groups as they fill up, the rest from the engine callback
        from masterthesis_amd import _lib as L
        d = L.ConvDesc(L.MT_BF16, 0, 8, 32, 32, 256, 256, 3, 3, 1, 1, L.PAD_REFLECT, 0, L.ACT_RELU, 0.01)
        gmax = int(L.load().mt_conv_bwd_weight_group_max(C.byref(d)))
        assert gmax >= 4
        assert res[False][2] == [] and sum(res[True][2]) == 6 and max(res[True][2]) <= gmax and len(res[True][2]) <= 2, \
            (res[False][2], res[True][2], gmax)
        assert res[True][3] == sorted(id(w) for w in ws) == res[False][3]
        assert torch.equal(res[True][1], res[False][1])
        for a, b in zip(res[True][0], res[False][0]):
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-5
    finally:
        ops.set_wgrad_group(True)
        ops.set_fused_grad_accumulation(fused_before)
        for w in ws:
            ops.set_grad_ready_hook(w, None)


def test_uses_of_one_weight_share_one_slab_sum(hip_device):
    """One weight applied at three scales (the MultiScaleDiscriminator pattern, reference networks.py:330-365): with the queue on,
    the three weight-gradient GEMMs write their slabs into one workspace and ONE slab sum adds them to param.grad."""
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(4)
    w = (torch.randn(128, 64, 4, 4) * 0.05).to(hip_device).requires_grad_()
    xs = [_rnd(4, 64, s, s, seed=30 + s).to(hip_device) for s in (64, 32, 16)]
    res = {}
    fused_before = ops._FUSE_WGRAD_ACC[0]
    ops.set_fused_grad_accumulation(True)
    try:
        for on in (False, True):
            ops.set_wgrad_group(on)
            fired = []
            w.grad = torch.zeros_like(w)
            ops.set_grad_ready_hook(w, lambda p, fired=fired: fired.append(1))
            ys = [ops.conv2d(x, w, None, stride=2, pad=1, pad_mode="zero", act="lrelu") for x in xs]
            loss = sum((y.float() ** 2).mean() for y in ys)
            ops.oplog_start()
            loss.backward()
            ev = ops.oplog_stop()
            res[on] = (w.grad.clone(), [k for k, d, ms in ev if k.startswith("wgrad")], len(fired))
        assert res[False][1] == ["wgrad"] * 3 and res[True][1] == ["wgrad"] * 3 + ["wgrad_sum"], (res[False][1], res[True][1])
        assert res[False][2] == 1 and res[True][2] == 1
        a, b = res[True][0], res[False][0]
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-6
    finally:
        ops.set_wgrad_group(True)
        ops.set_fused_grad_accumulation(fused_before)
        ops.set_grad_ready_hook(w, None)
