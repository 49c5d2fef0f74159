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


def test_batched_slab_sums_equal_single_finishes(hip_device):
    """Round 4: mt_conv_bwd_weight_finish_multi (the slab sums of a whole backward pass in one launch per 64 weights) against
    mt_conv_bwd_weight_finish per weight on the SAME slabs: bit-identical (same sums in the same order), accumulate on and off, for
    the generic slab form (3x3, 4x4 stride 2, transposed, odd channel counts) and for the layers that keep their own finish (7x7
    stem, thin 1x1) mixed into one call."""
    from masterthesis_amd import hip_ops as ops, _lib as L
    ops.set_compute_dtype(torch.bfloat16)
    lib = L.load()
    # transposed, N, Ci, H, W, Co, k, stride, pad, pad_mode, out_pad
    shapes = [(0, 4, 64, 32, 32, 128, 3, 1, 1, L.PAD_REFLECT, 0), (0, 8, 128, 16, 16, 256, 4, 2, 1, L.PAD_ZERO, 0),
              (1, 4, 128, 16, 16, 64, 3, 2, 1, L.PAD_ZERO, 1), (0, 2, 24, 20, 20, 40, 3, 2, 1, L.PAD_ZERO, 0),
              (0, 2, 3, 64, 64, 64, 7, 1, 3, L.PAD_REFLECT, 0), (1, 2, 64, 32, 32, 3, 1, 1, 0, L.PAD_ZERO, 0),
              (0, 4, 256, 8, 8, 512, 4, 2, 1, L.PAD_ZERO, 0)]
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr())
    descs, wss, nsl, shapes_w = [], [], [], []
    for i, (tr, N, Ci, H, W, Co, k, s_, pd, pm, op) in enumerate(shapes):
        d = L.ConvDesc(L.MT_BF16, tr, N, H, W, Ci, Co, k, k, s_, pd, pm, op, L.ACT_NONE, 0.0)
        ho, wo = C.c_int(), C.c_int()
        L.check(lib.mt_conv_out_hw(C.byref(d), C.byref(ho), C.byref(wo)), "out_hw")
        x = ops.canon(_rnd(N, Ci, H, W, seed=50 + i).to(hip_device))
        dy = ops.canon(_rnd(N, Co, ho.value, wo.value, seed=70 + i, scale=0.5).to(hip_device))
        nws = int(lib.mt_conv_bwd_weight_ws_bytes(C.byref(d)))
        ws = torch.empty((max(nws, 16),), dtype=torch.uint8, device=hip_device)
        ns = C.c_int(0)
        L.check(lib.mt_conv_bwd_weight_partial(C.byref(d), P(x), P(dy), None, P(ws), nws, 0, 1, C.byref(ns), st), "partial")
        assert ns.value > 0
        descs.append(d); wss.append(ws); nsl.append(ns.value)
        shapes_w.append((Ci, Co, k, k) if tr else (Co, Ci, k, k))
    n = len(descs)
    for accumulate in (0, 1):
        single = [torch.full(sw, 0.125, dtype=torch.float32, device=hip_device) for sw in shapes_w]
        multi = [t.clone() for t in single]
        for d, ws, ns, dw in zip(descs, wss, nsl, single):
            L.check(lib.mt_conv_bwd_weight_finish(C.byref(d), P(ws), ns, P(dw), accumulate, st), "finish")
        da = (L.ConvDesc * n)(*descs)
        wa = (C.c_void_p * n)(*[t.data_ptr() for t in wss])
        na = (C.c_int * n)(*nsl)
        ga = (C.c_void_p * n)(*[t.data_ptr() for t in multi])
        L.check(lib.mt_conv_bwd_weight_finish_multi(n, da, wa, na, ga, accumulate, st), "finish_multi")
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(multi, single)):
            assert torch.equal(a, b), (accumulate, i, (a - b).abs().max().item())
            assert a.abs().max().item() > 0.2


def test_slab_sums_wait_for_the_end_of_the_backward_pass(hip_device):
    """hip_ops parks the slab sums of a backward pass and launches them from the end-of-pass callback: same param.grad as with
    every sum behind its GEMM, and the op log shows one batched sum instead of one per layer."""
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(torch.bfloat16)
    ops.set_fused_grad_accumulation(True)
    try:
        ws = [(_rnd(32, 16, 3, 3, seed=1, scale=0.1), 1), (_rnd(48, 32, 4, 4, seed=2, scale=0.1), 2), (_rnd(64, 48, 3, 3, seed=3, scale=0.1), 1)]
        x0 = _rnd(4, 16, 32, 32, seed=9)
        grads = {}
        for on in (True, False):
            ops.set_wgrad_finish_batch(on)
            params = [torch.nn.Parameter(w.clone().to(hip_device)) for w, _ in ws]
            for p in params:
                p.grad = torch.zeros_like(p)
            h = x0.to(hip_device)
            ops.oplog_start()
            for p, (_, s_) in zip(params, ws):
                h = ops.conv2d(h, p, None, stride=s_, pad=1, pad_mode="zero", act="lrelu")
            h.float().sum().backward()
            log = ops.oplog_stop()
            sums = [d for k, d, _ in log if k == "wgrad_sum"]
            assert len(sums) == (1 if on else 0), (on, len(sums))
            if on:
                assert sums[0][-1] == 3          # three weights in the one batched launch
            grads[on] = [p.grad.clone() for p in params]
        for a, b in zip(grads[True], grads[False]):
            assert torch.equal(a, b)
    finally:
        ops.set_wgrad_finish_batch(True)
        ops.set_fused_grad_accumulation(False)
