"""Full-size checks (BASELINE config 2: 16 images of 256x256 per step, dim 64) through size-independent properties --
a CPU oracle of these shapes would take minutes, so the convolution family is checked through its adjoint
identities and the batched calls through batch-split invariance:

  <conv(x, w), g>  ==  <x, conv_bwd_data(g)>  ==  <w, conv_bwd_weight(g)>        (conv is bilinear in (x, w))

which exercises, at the sizes the bench runs, the 256x256 / 128x512 ping-pong kernels, the interior+ring data
gradient, the 256x256 weight-gradient kernel + unpack, split-K and the streaming class-head kernel."""
import zlib

import pytest
import torch

pytestmark = pytest.mark.gpu

# name, transposed, N, Ci, H, W, Co, k, stride, pad, pad_mode, out_pad
SHAPES = [
    ("K1_res3x3_256", False, 16, 256, 64, 64, 256, 3, 1, 1, "reflect", 0),
    ("K1_res3x3_256_N32", False, 32, 256, 64, 64, 256, 3, 1, 1, "reflect", 0),
    ("K1_res3x3_256_N64", False, 64, 256, 64, 64, 256, 3, 1, 1, "reflect", 0),     # config 3 (B=16): two-round wgrad
    ("stem7x7_3_64", False, 16, 3, 256, 256, 64, 7, 1, 3, "reflect", 0),
    ("down3x3s2_64_128", False, 16, 64, 256, 256, 128, 3, 2, 1, "reflect", 0),
    ("down3x3s2_128_256", False, 16, 128, 128, 128, 256, 3, 2, 1, "reflect", 0),
    ("up3x3s2_256_128", True, 32, 256, 64, 64, 128, 3, 2, 1, "zero", 1),
    ("up3x3s2_128_64", True, 16, 128, 128, 128, 64, 3, 2, 1, "zero", 1),
    ("torgb1x1_64_3", True, 16, 64, 256, 256, 3, 1, 1, 0, "zero", 0),
    ("dis3x3s2_512_1024", False, 32, 512, 16, 16, 1024, 3, 2, 1, "reflect", 0),
    ("dis3x3s2_1024_1024", False, 32, 1024, 8, 8, 1024, 3, 2, 1, "reflect", 0),
    ("msd4x4s2_1024_2048", False, 32, 1024, 8, 8, 2048, 4, 2, 1, "zero", 0),
    ("cls4x4_1024_2", False, 32, 1024, 4, 4, 2, 4, 1, 0, "zero", 0),
]


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_conv_adjoint_identities(shape, dtype, hip_device):
    from masterthesis_amd import hip_ops as ops
    ops.set_compute_dtype(dtype)
    name, tr, N, Ci, H, W, Co, k, stride, pad, pad_mode, out_pad = shape
    g = torch.Generator(device="cpu").manual_seed(zlib.crc32(name.encode()))   # fixed per shape (hash() of a string is per process)

    def rnd(*s, scale=1.0):
        t = torch.randn(*s, generator=g) * scale
        return (t.bfloat16().float() if dtype == torch.bfloat16 else t).to(hip_device)
    x = rnd(N, Ci, H, W).requires_grad_()
    wshape = (Ci, Co, k, k) if tr else (Co, Ci, k, k)
    w = rnd(*wshape, scale=(Ci * k * k) ** -0.5).requires_grad_()
    if tr:
        y = ops.conv_transpose2d(x, w, None, stride=stride, pad=pad, out_pad=out_pad)
    else:
        y = ops.conv2d(x, w, None, stride=stride, pad=pad, pad_mode=pad_mode)
    gy = rnd(*y.shape)
    y.backward(gy)
    s0 = _dot(y.detach(), gy)
    s1 = _dot(x.detach(), x.grad)
    s2 = _dot(w.detach(), w.grad)
    # scale of the comparison: |<y, g>| <= |y| |g|
    ref = (y.detach().double().norm() * gy.double().norm()).item()
    tol = (3e-5 if dtype == torch.float32 else 4e-3) * ref
    assert abs(s0 - s1) <= tol, f"{name}: <y,g>={s0:.6e} vs <x,dx>={s1:.6e} (tol {tol:.2e})"
    assert abs(s0 - s2) <= tol, f"{name}: <y,g>={s0:.6e} vs <w,dw>={s2:.6e} (tol {tol:.2e})"


@pytest.mark.parametrize("dtype,otol,gtol", [(torch.float32, 1e-4, 5e-3), (torch.bfloat16, 2e-2, 1.5e-1)], ids=["fp32", "bf16"])
def test_decoder_batch_split_invariance_fullsize(dtype, otol, gtol, hip_device):
    """The step batches the four translations of a phase into one decoder call: at full size the batched call must
    give what the separate calls give (per-sample AdaIN / LayerNorm), forward and parameter gradients.  fp32 pins the
    logic (outputs 1e-4, gradients 5e-3: the cancellation noise of the step test); in bf16 the two paths use different
    tilings, the activations differ in the last bit per layer and the normalisation backward amplifies that."""
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd.models.core import networks as Nn
    from masterthesis_amd.models.core.functions import init_weights
    ops.set_compute_dtype(dtype)     # (bf16: the two paths use different tilings -> last-bit rounding per layer)
    torch.manual_seed(0)
    dec = Nn.AdaINDecoder(3, dim=256, num_domains=2, latent_dim=8)
    init_weights(dec, "normal", 0.02)
    dec = dec.to(hip_device)
    B = 8
    zc = ops.canon(torch.randn(2 * B, 256, 64, 64, device=hip_device) * 0.5)
    zs = torch.randn(2 * B, 8, device=hip_device)
    cls = torch.zeros(2 * B, 2, device=hip_device)
    cls[:B, 0] = 1
    cls[B:, 1] = 1
    gy = torch.randn(2 * B, 3, 256, 256, device=hip_device)

    def run(chunks):
        for p in dec.parameters():
            p.grad = None
        outs = []
        for lo, hi in chunks:
            o = dec(zc[lo:hi], zs[lo:hi], cls[lo:hi])
            o.backward(ops.canon(gy[lo:hi]))
            outs.append(ops.to_nchw_f32(o).detach())
        return torch.cat(outs), [p.grad.detach().clone() for p in dec.parameters()]
    y1, g1 = run([(0, 2 * B)])
    y2, g2 = run([(0, B), (B, 2 * B)])
    rel = ((y1 - y2).norm() / y2.norm()).item()
    assert rel < otol, f"decoder outputs differ between one batched call and two calls: rel L2 {rel:.2e}"
    num = sum(((a - b).double().norm() ** 2).item() for a, b in zip(g1, g2)) ** 0.5
    den = sum((b.double().norm() ** 2).item() for b in g2) ** 0.5
    assert num / den < gtol, f"decoder parameter gradients differ: rel L2 {num / den:.2e}"


FULL_SIZE_CONFIGS = {
    # BASELINE.json configs[1]: 2 domains, 256x256, batch 8
    "configs1_single_scale": dict(num_domains=2, batch_size=8, crop_size=256, ms_dis=False),
    "configs1_multi_scale": dict(num_domains=2, batch_size=8, crop_size=256, ms_dis=True),
    # configs[2] (and the per-GPU share of configs[3]): 4 domains, 256x256, batch 16, multi-scale discriminators
    "configs2_d4_b16_multi_scale": dict(num_domains=4, batch_size=16, crop_size=256, ms_dis=True),
    # the per-GPU workload shape of configs[4]: 4 domains, 512x512 (batch 2 pairs)
    "configs4_d4_512_multi_scale": dict(num_domains=4, batch_size=2, crop_size=512, ms_dis=True),
}


@pytest.mark.parametrize("cfg", list(FULL_SIZE_CONFIGS))
def test_full_size_step_runs_and_is_finite(cfg, tmp_path, hip_device):
    """Two full-size steps in bf16 at the sizes BASELINE.json names: every loss finite and in the range of a GAN at
    initialisation, parameters of every network move, no NaN anywhere."""
    import argparse
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from masterthesis_amd import models
    from masterthesis_amd.dataset import SyntheticDataset
    o = argparse.Namespace(precision="bf16", **FULL_SIZE_CONFIGS[cfg])
    args = bench.model_args(o, str(tmp_path))
    torch.manual_seed(0)
    M = models.AdaINModel(args)
    M.initialize()
    ds = SyntheticDataset(args, length=8, seed=7)
    items = [ds[i % 8] for i in range(o.batch_size)]
    batch = {k: torch.stack([it[k] for it in items]).to(hip_device) for k in items[0]}
    before = {n: torch.cat([p.detach().flatten()[:1000].float() for p in M.model[n].parameters()]).clone() for n in M.model}
    for it in range(2):
        M.update_lr()
        M.set_inputs(batch)
        M.optimize_parameters(it)
    losses = M.sync_losses()
    for k, v in losses.items():
        assert v == v and abs(v) < 1e6, f"loss {k} = {v}"
    # at initialisation (N(0, 0.02) weights) every logit is ~0: BCE terms sit at ln 2 per term
    import math
    n_scales = 3 if o.ms_dis else 1
    assert abs(losses["d_adv"] - 2 * n_scales * math.log(2)) < 0.05 * n_scales, losses["d_adv"]
    assert abs(losses["g_cls"] - 5 * n_scales * math.log(2)) < 0.2 * n_scales, losses["g_cls"]
    for n in M.model:
        after = torch.cat([p.detach().flatten()[:1000].float() for p in M.model[n].parameters()])
        assert torch.isfinite(after).all(), f"{n} has non-finite parameters"
        assert (after - before[n]).abs().max().item() > 0, f"{n} did not move"


@pytest.mark.parametrize("deterministic", [False, True], ids=["atomics", "deterministic"])
def test_short_training_run_bf16_tracks_fp32(deterministic, tmp_path, hip_device):
    """80 optimisation steps (128x128, 4 pairs, dim 64) in bf16 and in fp32 from the same seed: the reconstruction
    losses must go down and the bf16 trajectory must stay within a few percent of the fp32 one; also with the
    fixed-order reductions of deterministic mode (separate statistics pass instead of the fused epilogue) at real widths."""
    from masterthesis_amd import hip_ops as _ops
    from masterthesis_amd.models.core import misc as _misc
    _ops.set_deterministic(deterministic)
    _misc.set_random_source(None)
    try:
        _short_training_run(deterministic, tmp_path, hip_device)
    finally:
        _ops.set_deterministic(False)


def _short_training_run(deterministic, tmp_path, hip_device):
    import argparse
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from masterthesis_amd import models
    from masterthesis_amd.dataset import SyntheticDataset
    out = {}
    for prec in ("bf16", "fp32"):
        o = argparse.Namespace(precision=prec, num_domains=2, batch_size=4, crop_size=128, ms_dis=False)
        args = bench.model_args(o, str(tmp_path / prec))
        torch.manual_seed(0)
        M = models.AdaINModel(args)
        M.initialize()
        ds = SyntheticDataset(args, length=16, seed=5)
        batches = []
        for b in range(4):
            items = [ds[b * 4 + i] for i in range(4)]
            batches.append({k: torch.stack([it[k] for it in items]).to(hip_device) for k in items[0]})
        first = None
        for it in range(80):
            M.update_lr()
            M.set_inputs(batches[it % 4])
            M.optimize_parameters(it)
            if it == 3:
                first = dict(M.sync_losses())
        out[prec] = (first, dict(M.sync_losses()))
        for n in M.model:
            assert all(torch.isfinite(p).all() for p in M.model[n].parameters()), f"{prec}: {n} has non-finite weights"
    for prec, (first, last) in out.items():
        assert last["l1_self_rec"] < first["l1_self_rec"] and last["total_g"] < first["total_g"], (prec, first, last)
    for k in ("total_g", "l1_self_rec", "l1_cc_rec"):
        a, b = out["bf16"][1][k], out["fp32"][1][k]
        assert abs(a - b) <= 0.10 * abs(b) + 0.05, f"{k}: bf16 {a} vs fp32 {b} after 80 steps"
    # the discriminator loss is the volatile one: around step 80 the discriminators start to separate real from fake
    # and the moment differs from run to run (atomics order) as much as between precisions -- 1.27 vs 1.79 was seen
    # once in ~10 runs -- so it only has to stay in the range of a working GAN (2 ln 2 + ln 2 = 2.08 at initialisation)
    for prec in ("bf16", "fp32"):
        d = out[prec][1]["d_total"]
        assert 0.3 < d < 2.5, f"{prec}: d_total {d} after 80 steps"
    print(f"MT_DIAG short run deterministic={deterministic}: " +
          ", ".join(f"{p} d_total {out[p][1]['d_total']:.4f} total_g {out[p][1]['total_g']:.4f}" for p in ("bf16", "fp32")))
    # (deterministic mode makes each run bit-reproducible -- 1.1458 / 1.5657 for bf16 / fp32 in two consecutive runs --
    # but does not bring the two PRECISIONS closer: when the discriminators start to separate real from fake is chaotic
    # in the rounding, so d_total keeps its range check; the pin for deterministic mode is bit-identity of two runs,
    # tests/test_graph_gpu.py::test_deterministic_mode_is_bit_reproducible)


def test_sampling_path_at_deployment_size(hip_device):
    """sample.py's forward_random / forward_reference at 540x960 (sample.py:79-91), dim 64: bf16 output within bf16
    rounding of the fp32 path of the same weights and draws; odd intermediate sizes (135 -> 67 in the style encoder's
    average pools) and the 2x transposed convolutions of the decoder at 135x240 -> 540x960."""
    import argparse
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd import models
    from masterthesis_amd.models.core import misc
    outs = {}
    img = torch.rand(1, 3, 540, 960, generator=torch.Generator().manual_seed(3)).to(hip_device) * 2 - 1
    c = torch.eye(4, device=hip_device)[[2]]
    z = torch.randn(1, 8, generator=torch.Generator().manual_seed(4)).to(hip_device)
    for prec in ("fp32", "bf16"):
        a = argparse.Namespace(mode="test", precision=prec, input_dim=3, dim=64, enc_norm="instance", num_domains=4,
                               latent_dim=8, up_type="transpose", dec_norm="layer", use_dropout=False, init_type="normal",
                               init_gain=0.02, resume=None, gpu_ids=[0], batch_size=1, concat=False, reparam=False)
        torch.manual_seed(0)
        M = models.AdaINModel(a)
        M.initialize()
        for net in M.model:
            M.model[net].eval()             # no content noise: the two precisions see the same function
        with torch.no_grad():
            r, _, _ = M.forward_random(img, z, c)
            f, _, _ = M.forward_reference(img, img.flip(3), c)
        outs[prec] = (ops.to_nchw_f32(r), ops.to_nchw_f32(f))
    for a, b in zip(outs["bf16"], outs["fp32"]):
        assert a.shape == (1, 3, 540, 960) and torch.isfinite(a).all()
        rel = ((a - b).norm() / b.norm()).item()
        assert rel < 3e-2, f"bf16 sampling output differs from fp32 by rel L2 {rel:.2e}"
