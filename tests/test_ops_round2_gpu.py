"""Op-level GPU tests of the C-ABI entry points added in round 2, each against plain torch on the CPU (or against
the two-launch path it replaces): one-launch loss expression, device-record Adam, device-state Philox draws, the
two-stage reproducible statistics, finalize+apply in one launch, activation-derivative + bias gradient in one pass."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from masterthesis_amd import _lib as L
    return L, L.load()


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def test_loss_sum_matches_torch_expression(hip_device):
    from masterthesis_amd import hip_ops as ops
    g = torch.Generator().manual_seed(0)
    vals = torch.randn(7, generator=g)
    ts = [v.clone().to(hip_device).requires_grad_() for v in vals]
    groups = [("adv", [(ts[0], 1.0), (ts[1], 1.0), (ts[2], 0.5)], 1.0, 1.0),
              ("cls", [(ts[3], 5.0)], 2.0, 2.0),
              ("rec", [(ts[4], 10.0), (ts[5], 10.0)], 1.0, 1.0),
              ("kl", [(ts[6], 0.01)], 8.0, 1.0)]          # differentiated with weight 8 (world size), reported with 1
    total, v, reported = ops.loss_sum(groups)
    (total * 3.0).backward()
    ref = [t.detach().cpu().double().requires_grad_() for t in ts]
    adv = ref[0] + ref[1] + 0.5 * ref[2]
    cls, rec, kl = 5.0 * ref[3], 10.0 * ref[4] + 10.0 * ref[5], 0.01 * ref[6]
    want_total = adv + 2.0 * cls + rec + 8.0 * kl
    (want_total * 3.0).backward()
    assert abs(total.item() - want_total.item()) < 1e-5 * max(1.0, abs(want_total.item()))
    assert abs(reported.item() - (adv + 2.0 * cls + rec + kl).item()) < 1e-5 * max(1.0, abs(want_total.item()))
    for name, want in (("adv", adv), ("cls", cls), ("rec", rec), ("kl", kl)):
        assert abs(v[name].item() - want.item()) < 1e-5 * max(1.0, abs(want.item())), name
    for a, b in zip(ts, ref):
        assert abs(a.grad.item() - b.grad.item()) < 1e-6 * max(1.0, abs(b.grad.item()))


def test_fused_adam_device_record_matches_torch_adam(hip_device):
    """FusedAdam (step count / bias corrections / lr in a device record, gradients cleared by the update) against
    torch.optim.Adam on the CPU over several steps, with a learning-rate change in between."""
    from masterthesis_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(7, 5, 3, 3), (7,), (11, 13)]
    p0 = [torch.randn(s, generator=g) for s in shapes]
    ref_p = [torch.nn.Parameter(p.clone()) for p in p0]
    dev_p = [torch.nn.Parameter(p.clone().to(hip_device)) for p in p0]
    ref = torch.optim.Adam(ref_p, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    opt = FusedAdam(dev_p, lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-4)
    for it in range(6):
        if it == 3:
            for o in (ref, opt):
                o.param_groups[0]["lr"] = 2.5e-4
        opt.zero_grad()
        grads = [torch.randn(s, generator=g) for s in shapes]
        for rp, dp, gr in zip(ref_p, dev_p, grads):
            rp.grad = gr.clone()
            dp.grad.copy_(gr.to(hip_device))
        ref.step()
        opt.step()
        assert float(opt.flat_grad().abs().max()) == 0.0, "the update must clear the gradients it consumed"
    for rp, dp in zip(ref_p, dev_p):
        err = (dp.detach().cpu() - rp.detach()).abs().max().item()
        assert err < 2e-6, err
    assert int(opt._dev.view(torch.int32)[1].item()) == 6 == opt._step_count_mt


def test_device_state_noise_is_standard_normal_and_advances(hip_device):
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(torch.float32)
    st = torch.tensor([12345, 0], dtype=torch.int64).to(hip_device)
    x = ops.canon(torch.zeros(2, 64, 64, 64, device=hip_device))
    a = ops.gaussian_noise_add_dev(x, st)
    b = ops.gaussian_noise_add_dev(x, st)
    assert st.tolist() == [12345, 2]
    fa, fb = ops.to_nchw_f32(a).flatten(), ops.to_nchw_f32(b).flatten()
    assert abs(fa.mean().item()) < 5e-3 and abs(fa.std().item() - 1.0) < 5e-3
    assert (fa - fb).abs().max().item() > 1.0, "consecutive draws must differ"
    # the same (seed, counter) reproduces the draw
    st2 = torch.tensor([12345, 0], dtype=torch.int64).to(hip_device)
    assert torch.equal(ops.to_nchw_f32(ops.gaussian_noise_add_dev(x, st2)).flatten(), fa)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 8, 5, 7), (2, 24, 33, 17), (2, 256, 16, 16), (1, 520, 6, 6), (4, 64, 64, 64)])
def test_two_stage_statistics_are_exact_and_reproducible(shape, dtype, hip_device):
    from masterthesis_amd import hip_ops as ops
    L, lib = _lib()
    ops.set_compute_dtype(dtype)
    N, Cc, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    x0 = torch.randn(shape, generator=g)
    x = ops.canon(x0.to(hip_device))
    xr = ops.to_nchw_f32(x).cpu().double()              # (what the kernel actually sees after bf16 rounding)
    Cp, HW = ops.padc(Cc), H * W
    mt = L.MT_BF16 if dtype == torch.bfloat16 else L.MT_F32
    nparts = int(lib.mt_nc_stats_parts(mt, N, HW, Cp))
    assert 1 <= nparts <= 64
    outs = []
    for _ in range(2):
        part = torch.full((N, nparts, Cp, 2), float("nan"), dtype=torch.float32, device=hip_device)
        L.check(lib.mt_nc_stats(mt, P(x), P(part), N, HW, Cp, S()), "mt_nc_stats")
        outs.append(part.clone())
    assert torch.equal(outs[0], outs[1]), "statistics differ between two identical launches"
    tot = outs[0].double().sum(1).cpu()
    assert torch.isfinite(tot).all()
    s1, s2 = xr.sum((2, 3)), (xr * xr).sum((2, 3))
    assert (tot[:, :Cc, 0] - s1).abs().max().item() <= 1e-4 * (s1.abs().max().item() + 1.0)
    assert (tot[:, :Cc, 1] - s2).abs().max().item() <= 1e-4 * s2.abs().max().item()
    assert float(tot[:, Cc:].abs().max()) == 0.0 if Cp > Cc else True


@pytest.mark.parametrize("mode_name", ["instance", "adain", "layer"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_norm_apply_fused_equals_finalize_plus_apply(mode_name, dtype, hip_device):
    from masterthesis_amd import hip_ops as ops
    L, lib = _lib()
    ops.set_compute_dtype(dtype)
    mode = {"instance": L.NORM_INSTANCE, "adain": L.NORM_ADAIN, "layer": L.NORM_LAYER}[mode_name]
    N, Cc, H, W = 3, 20, 9, 11
    g = torch.Generator().manual_seed(7)
    x = ops.canon(torch.randn(N, Cc, H, W, generator=g).to(hip_device))
    res = ops.canon(torch.randn(N, Cc, H, W, generator=g).to(hip_device))
    gb = torch.randn(N, 2 * Cc, generator=g).to(hip_device) if mode == L.NORM_ADAIN else None
    gamma = torch.randn(Cc, generator=g).to(hip_device) if mode == L.NORM_LAYER else None
    beta = torch.randn(Cc, generator=g).to(hip_device) if mode == L.NORM_LAYER else None
    Cp, HW = ops.padc(Cc), H * W
    mt = L.MT_BF16 if dtype == torch.bfloat16 else L.MT_F32
    # complete per-image sums [N][Cp][2] (what the convolution's epilogue hands over)
    xr = ops.to_nchw_f32(x)
    sums = torch.zeros(N, Cp, 2, device=hip_device)
    sums[:, :Cc, 0], sums[:, :Cc, 1] = xr.sum((2, 3)), (xr * xr).sum((2, 3))
    coef_a = torch.empty(4, N, Cp, device=hip_device)
    ya = ops.new_act(N, Cc, H, W, x.dtype, hip_device)
    L.check(lib.mt_norm_finalize(mode, P(sums), P(gb), P(gamma), P(beta), P(coef_a[0]), P(coef_a[1]), P(coef_a[2]), P(coef_a[3]),
                                 N, HW, Cc, Cp, 1e-5, 1, S()), "mt_norm_finalize")
    L.check(lib.mt_scale_shift_act(mt, P(x), P(coef_a[0]), P(coef_a[1]), P(res), P(ya), N, HW, Cp, L.ACT_RELU, 0.01, S()),
            "mt_scale_shift_act")
    coef_b = torch.full((4, N, Cp), float("nan"), device=hip_device)
    yb = ops.new_act(N, Cc, H, W, x.dtype, hip_device)
    L.check(lib.mt_norm_apply_fused(mt, mode, P(x), P(sums), P(gb), P(gamma), P(beta), P(res), P(yb), P(coef_b), N, HW, Cc, Cp,
                                    L.ACT_RELU, 0.01, 1e-5, S()), "mt_norm_apply_fused")
    assert torch.equal(coef_a, coef_b)
    assert torch.equal(ops.to_nchw_f32(ya), ops.to_nchw_f32(yb))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_act_bwd_bias_equals_act_bwd_then_column_sum(dtype, hip_device):
    from masterthesis_amd import hip_ops as ops
    L, lib = _lib()
    ops.set_compute_dtype(dtype)
    N, Cc, H, W = 2, 20, 13, 9
    g = torch.Generator().manual_seed(3)
    dy = ops.canon(torch.randn(N, Cc, H, W, generator=g).to(hip_device))
    y = ops.canon(torch.randn(N, Cc, H, W, generator=g).to(hip_device))
    Cp, npix = ops.padc(Cc), N * H * W
    mt = L.MT_BF16 if dtype == torch.bfloat16 else L.MT_F32
    dz_a = ops.new_act(N, Cc, H, W, dy.dtype, hip_device, zero=True)
    L.check(lib.mt_act_bwd(mt, P(dy), P(y), P(dz_a), npix * Cp, L.ACT_LRELU, 0.01, S()), "mt_act_bwd")
    nws = int(lib.mt_act_bwd_bias_ws_bytes(Cp))
    ws = torch.empty(nws, dtype=torch.uint8, device=hip_device)
    dz_b = ops.new_act(N, Cc, H, W, dy.dtype, hip_device, zero=True)
    db = torch.full((Cc,), 2.0, device=hip_device)                       # accumulate onto existing contents
    L.check(lib.mt_act_bwd_bias(mt, P(dy), P(y), P(dz_b), npix, Cp, Cc, L.ACT_LRELU, 0.01, P(db), 1, P(ws), nws, S()),
            "mt_act_bwd_bias")
    assert torch.equal(ops.to_nchw_f32(dz_a), ops.to_nchw_f32(dz_b))
    want = ops.to_nchw_f32(dz_a).double().sum((0, 2, 3)) + 2.0
    assert (db.double() - want).abs().max().item() < 1e-4 * (want.abs().max().item() + 1.0)


@pytest.mark.parametrize("fused", [False, True], ids=["autograd_grads", "accumulate_into_param_grad"])
def test_linear_grouped_matches_separate_layers(fused, hip_device):
    from masterthesis_amd import hip_ops as ops
    g = torch.Generator().manual_seed(11)
    n, i, o, G = 5, 24, 40, 4
    x0 = torch.randn(n, i, generator=g)
    W = [torch.randn(o, i, generator=g) * 0.1 for _ in range(G)]
    Bv = [torch.randn(o, generator=g) for _ in range(G)]
    cot = [torch.randn(n, o, generator=g) for _ in range(G)]
    xr = x0.clone().double().requires_grad_()
    Wr = [w.clone().double().requires_grad_() for w in W]
    Br = [b.clone().double().requires_grad_() for b in Bv]
    sum(((xr @ Wr[k].t() + Br[k]) * cot[k].double()).sum() for k in range(G)).backward()
    xd = x0.clone().to(hip_device).requires_grad_()
    Wd = [torch.nn.Parameter(w.clone().to(hip_device)) for w in W]
    Bd = [torch.nn.Parameter(b.clone().to(hip_device)) for b in Bv]
    ops.set_fused_grad_accumulation(fused)
    try:
        if fused:
            for p in Wd + Bd:
                p.grad = torch.ones_like(p)             # the backward ADDS to what is there
        ys = ops.linear_grouped(xd, list(zip(Wd, Bd)))
        for k in range(G):
            want = (x0 @ W[k].t() + Bv[k])
            assert (ys[k].detach().cpu() - want).abs().max().item() < 1e-4
        torch.autograd.backward(ys, [c.to(hip_device) for c in cot])
    finally:
        ops.set_fused_grad_accumulation(True)
    off = 1.0 if fused else 0.0
    assert (xd.grad.cpu().double() - xr.grad).abs().max().item() < 1e-4
    for k in range(G):
        assert (Wd[k].grad.cpu().double() - off - Wr[k].grad).abs().max().item() < 1e-4, k
        assert (Bd[k].grad.cpu().double() - off - Br[k].grad).abs().max().item() < 1e-4, k


PACK_DESCS = [
    # transposed, N, H, W, Ci, Co, k, stride, pad, pad_mode, out_pad
    (0, 2, 16, 16, 256, 256, 3, 1, 1, 1, 0),        # K1: forward image + ring data-gradient image (25 tap runs)
    (0, 2, 16, 16, 64, 128, 3, 2, 1, 1, 0),         # stride 2: four sub-pixel phase images for the data gradient
    (0, 2, 16, 16, 520, 1030, 4, 2, 1, 0, 0),       # 4x4, channel counts that are not multiples of the tiles
    (0, 2, 32, 32, 3, 64, 7, 1, 3, 1, 0),           # stem: 49 taps, 3 input channels
    (1, 2, 16, 16, 128, 64, 3, 2, 1, 0, 1),         # ConvTranspose2d (IOHW): "rows near" forward image
    (0, 2, 8, 8, 1024, 1024, 3, 2, 1, 1, 0),
    (0, 2, 8, 8, 256, 256, 1, 1, 0, 0, 0),          # 1x1
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", PACK_DESCS, ids=[f"{'T' if c[0] else 'C'}{c[4]}x{c[5]}k{c[6]}s{c[7]}" for c in PACK_DESCS])
def test_batched_weight_pack_equals_single_pack(case, dtype, hip_device):
    """mt_conv_pack_multi_* (LDS-staged tiles, one launch for many images) must produce byte-identical images to
    mt_conv_pack (one element per thread) for forward and data-gradient packs of every layout the step uses."""
    L, lib = _lib()
    tr, N, H, W, Ci, Co, k, st, pad, pm, op = case
    mt = L.MT_BF16 if dtype == torch.bfloat16 else L.MT_F32
    desc = L.ConvDesc(mt, tr, N, H, W, Ci, Co, k, k, st, pad, pm, op, L.ACT_NONE, 0.01)
    g = torch.Generator().manual_seed(Ci + Co + k)
    shape = (Ci, Co, k, k) if tr else (Co, Ci, k, k)
    w = torch.randn(shape, generator=g).to(hip_device)
    singles, packs = [], []
    for which in (L.PACK_FWD, L.PACK_BWD_DATA):
        nb = max(int(lib.mt_conv_pack_bytes(C.byref(desc), which)), 16)
        a = torch.full((nb,), 0x5A, dtype=torch.uint8, device=hip_device)
        L.check(lib.mt_conv_pack(C.byref(desc), which, P(w), P(a), S()), "mt_conv_pack")
        singles.append(a)
        packs.append(torch.full((nb,), 0xA5, dtype=torch.uint8, device=hip_device))
    n = 2
    descs = (L.ConvDesc * n)(desc, desc)
    whichs = (C.c_int * n)(L.PACK_FWD, L.PACK_BWD_DATA)
    ws = (C.c_void_p * n)(w.data_ptr(), w.data_ptr())
    pk = (C.c_void_p * n)(packs[0].data_ptr(), packs[1].data_ptr())
    host = C.create_string_buffer(int(lib.mt_conv_pack_multi_table_bytes(n)))
    ne, nblk = C.c_int(), C.c_int()
    L.check(lib.mt_conv_pack_multi_build(n, descs, whichs, ws, pk, host, C.byref(ne), C.byref(nblk)), "build")
    dev = torch.frombuffer(host, dtype=torch.uint8).clone().to(hip_device)
    L.check(lib.mt_conv_pack_multi_run(P(dev), ne.value, nblk.value, S()), "run")
    torch.cuda.synchronize()
    for which, a, b in zip(("fwd", "bwd_data"), singles, packs):
        assert torch.equal(a, b), f"{which}: batched pack differs in {(a != b).sum().item()} of {a.numel()} bytes"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch4s2_multi_is_the_unfold_of_a_4x4_stride2_conv(dtype, hip_device):
    """ops.patch4s2_multi: patches of nn.Conv2d(.., 4, 2, 1) of several inputs as one batch of 4x4 mini-images, so that the
    SAME weights as a 4x4 / stride 4 convolution reproduce the original outputs (the weight-shared scales of the reference's
    MultiScaleDiscriminator, networks.py:330-365); backward = the adjoint."""
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(dtype)
    try:
        g = torch.Generator().manual_seed(11)
        xs = [torch.randn(3, 16, s, t, generator=g).bfloat16().float() for s, t in ((8, 8), (4, 6), (2, 2))]
        w = (torch.randn(24, 16, 4, 4, generator=g) * 0.1).bfloat16().float()
        xd = [x.to(hip_device).requires_grad_() for x in xs]
        wd = w.to(hip_device).requires_grad_()
        col = ops.patch4s2_multi(xd)
        counts = [3 * (x.shape[2] // 2) * (x.shape[3] // 2) for x in xs]
        assert tuple(col.shape) == (sum(counts), 16, 4, 4) and ops.is_canonical(col)
        # the patches themselves: torch unfold of the zero-padded input
        ref_cols = []
        for x in xs:
            u = torch.nn.functional.unfold(x, 4, padding=1, stride=2)            # [N, C*16, L]
            ref_cols.append(u.transpose(1, 2).reshape(-1, 16, 4, 4))
        assert torch.equal(ops.to_nchw_f32(col).cpu(), torch.cat(ref_cols).to(torch.float32))
        y = ops.conv2d(col, wd, None, stride=4, pad=0, pad_mode="zero", act="lrelu")
        ys = ops.split_pixels(y, [(3, x.shape[2] // 2, x.shape[3] // 2) for x in xs])
        xr = [x.clone().requires_grad_() for x in xs]
        wr = w.clone().requires_grad_()
        yr = [torch.nn.functional.leaky_relu(torch.nn.functional.conv2d(x, wr, None, 2, 1), 0.01) for x in xr]
        gys = [torch.randn(*t.shape, generator=g).bfloat16().float() for t in yr]
        sum((a * b).sum() for a, b in zip(yr, gys)).backward()
        sum((a.float() * b.to(hip_device)).sum() for a, b in zip(ys, gys)).backward()
        tol = 2e-2 if dtype == torch.bfloat16 else 2e-4
        for a, b in zip(ys, yr):
            assert a.shape == b.shape and ops.is_canonical(a)
            assert (a.float().cpu() - b).norm() <= tol * b.norm() + 1e-5
        for a, b in zip(xd, xr):
            assert (a.grad.float().cpu() - b.grad).norm() <= tol * b.grad.norm() + 1e-5
        assert (wd.grad.cpu() - wr.grad).norm() <= tol * wr.grad.norm() + 1e-5
    finally:
        ops.set_compute_dtype(torch.bfloat16)


def test_multi_scale_discriminator_merged_layers_match_per_scale_layers(hip_device):
    """MultiScaleDiscriminator with its deep layers run on merged mini-image batches (the default) against the per-scale form:
    outputs and every parameter / input gradient agree (same products, sums in another order; bf16)."""
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd.models.core import networks as nets
    from masterthesis_amd.models.core.functions import init_weights
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(5)
    D = nets.MultiScaleDiscriminator(3, dim=16, n_layers=6, num_domains=2)
    init_weights(D, "normal", 0.05)
    D = D.to(hip_device)
    x0 = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(6)).to(hip_device)
    res = {}
    try:
        for on in (False, True):
            nets._MSD_MERGE[0] = on
            for p in D.parameters():
                p.grad = None
            x = x0.clone().requires_grad_()
            outs = D(x)
            merged = [nets.MultiScaleDiscriminator._mergeable(l, [torch.empty(4, l.block[l._ci].in_channels, 2 * s, 2 * s,
                                                                              device=hip_device) for s in (4, 2, 1)])
                      for l in D.model]
            loss = sum((d.float() ** 2).mean() + (c.float() ** 2).mean() for d, c in outs)
            loss.backward()
            res[on] = ([t.float().detach().clone() for o in outs for t in o], [p.grad.clone() for p in D.parameters()],
                       ops.to_nchw_f32(x.grad), merged)
        assert any(res[True][3]) and not any(res[False][3])
        for a, b in zip(res[True][0], res[False][0]):
            assert (a - b).norm() <= 3e-2 * b.norm() + 1e-4
        for a, b in zip(res[True][1], res[False][1]):
            assert (a - b).norm() <= 6e-2 * b.norm() + 1e-6
        assert (res[True][2] - res[False][2]).norm() <= 6e-2 * res[False][2].norm() + 1e-6
    finally:
        nets._MSD_MERGE[0] = True


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Co", [1, 2, 3, 5, 6, 7, 8])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_thin_head_weight_gradient_stays_inside_its_tensor(Co, accumulate, dtype, hip_device):
    """thin_wgrad_kernel is instantiated for 1 / 2 / 4 / 8 output rows and used to store all of them; with
    --num_domains 3, 5, 6, 7 (cls heads 2048 -> D, reference networks.py:452) the extra rows landed on the next parameter of
    the flat gradient buffer.  A canary region behind dw must stay untouched, in both accumulate modes, and the rows that are
    written must equal the fp32 reference."""
    import ctypes as C
    from masterthesis_amd import hip_ops as ops, _lib as L
    ops.set_compute_dtype(dtype)
    lib = L.load()
    N, Ci, H, W = 2, 512, 3, 3
    g = torch.Generator().manual_seed(100 + Co)
    x = torch.randn(N, Ci, H, W, generator=g).bfloat16().float()
    dy = torch.randn(N, Co, H, W, generator=g).bfloat16().float()
    desc = L.ConvDesc(L.MT_BF16 if dtype == torch.bfloat16 else L.MT_F32, 0, N, H, W, Ci, Co, 1, 1, 1, 0, L.PAD_ZERO, 0,
                      L.ACT_NONE, 0.0)
    xd, dyd = ops.canon(x.to(hip_device)), ops.canon(dy.to(hip_device))
    n = Co * Ci
    buf = torch.full((n + 8 * Ci,), 7.0, dtype=torch.float32, device=hip_device)      # dw followed by the canary
    start = torch.full((n,), 0.5, dtype=torch.float32, device=hip_device)
    buf[:n] = start
    nws = int(lib.mt_conv_bwd_weight_ws_bytes(C.byref(desc)))
    ws = torch.empty((max(nws, 16),), dtype=torch.uint8, device=hip_device)
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    L.check(lib.mt_conv_bwd_weight(C.byref(desc), P(xd), P(dyd), P(buf), None, P(ws), nws, accumulate, st), "thin wgrad")
    torch.cuda.synchronize()
    assert (buf[n:] == 7.0).all(), f"Co={Co}: {int((buf[n:] != 7.0).sum())} floats written behind dw"
    ref = torch.einsum("nohw,nihw->oi", dy, x)
    got = (buf[:n] - (start if accumulate else 0)).view(Co, Ci).cpu()
    assert ((got - ref).norm() / ref.norm()).item() < 2e-5
