"""GPU parity of the full G+D training step (through the C ABI) against the golden vectors recorded
from the reference: loss scalars, the gradients consumed by each of the 7 Adam steps, and the
post-step parameters.  fp32 path: the north-star tolerance (1e-3 relative); bf16 path: bf16 rounding."""
import os

import numpy as np
import pytest
import torch

from helpers import checksum, load_gold, product_args, sub

pytestmark = pytest.mark.gpu


def _build(name, tmp_path, precision):
    from masterthesis_amd import models
    from masterthesis_amd.models.core import misc
    z, meta = load_gold(name)
    args = product_args(meta["args"], str(tmp_path), precision)
    M = getattr(models, meta["model"])(args)
    M.initialize()
    for net in M.model:
        M.model[net].load_state_dict(sub(z, f"init/{net}"))
    return z, meta, M, misc


def _rel(a, b):
    a, b = torch.as_tensor(a).double().cpu().flatten(), torch.as_tensor(b).double().cpu().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _fp64_truth(z, meta, it_count):
    """The oracle in float64 with the recorded draws: ground truth for gradients.  (The reference's own
    fp32 CPU gradients carry up to several % of cancellation noise in the tiny-width fixtures, measured
    against this fp64 run, so they only pin the oracle -- tests/test_oracle_golden.py.)"""
    from oracle import step
    a = meta["args"]
    args = step.default_args(**{k: a[k] for k in vars(step.default_args()) if k in a})
    args.model = meta["model"]
    nets_present = sorted({k.split("/")[1] for k in z.files if k.startswith("init/")})
    O = step.OracleModel({n: sub(z, f"init/{n}") for n in nets_present}, args, dtype=torch.float64)
    batch = sub(z, "batch")
    out = []
    for it in range(it_count):
        rng = step.ReplayRng([z[f"rng/{it}/{i}"] for i in range(meta["rng_counts"][it])], dtype=torch.float64)
        seen = []
        for net, opt in O.opt.items():
            if not hasattr(opt, "_orig_step"):
                opt._orig_step = opt.step

            def hooked(_net=net, _opt=opt):
                seen.append((_net, {k: p.grad.detach().clone() for k, p in O.P[_net].items() if p.requires_grad}))
                return _opt._orig_step()
            opt.step = hooked
        O.update_lr()
        O.set_inputs(batch)
        O.optimize_parameters(it, rng)
        out.append((dict(O.loss), seen, O.state()))
    return out


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["adain_step_d2", "adain_step_d4_b2", "base_step_concat_reparam", "adain_step_lsgan",
                                  "adain_step_hinge", "adain_step_ragan", "adain_step_nearest", "adain_step_sn", "adain_step_dc", "base_step_concat", "adain_step_dropout",
                                  "base_step_concat_dropout", "adain_step_norms", "adain_step_bn", "adain_step_ms", "adain_step_wgangp"])
def test_training_step_matches_reference(name, precision, tmp_path, hip_device):
    z, meta, M, misc = _build(name, tmp_path, precision)
    torch.set_num_threads(8)
    truth = _fp64_truth(z, meta, meta["steps"])
    batch = sub(z, "batch")
    loss_tol = 1e-3 if precision == "fp32" else 3e-2
    # per backward phase (optimizer-step index): D1, D2 | Ec, Es, Dec (phase 3) | Ec, Dec (phase 4).
    # Phase-4 gradients at initialisation are cancellation dominated (tiny-width fixtures: even the
    # reference's fp32 CPU run is 3-8 % away from fp64 there), so bf16 storage only gets a sanity bound.
    if precision == "fp32":
        # Discriminator phases: smooth in the inputs, strict bound.  Generator phases: the L1 reconstruction terms
        # (gradient = sign(img - fake)) and the ReLU masks make the gradient piecewise constant in the forward
        # activations.  Measured on the float64 oracle with the inputs perturbed by 1e-6 relative (8 draws per fixture):
        # phase 3 moves by 3e-6 .. 1.5e-2, phase 4 (cancellation dominated at initialisation; the reference's own
        # fp32 run is 3-8 % from fp64) by 2e-5 .. 1.2e-1 -- so an fp32 run lands anywhere in that range from run to
        # run (atomics order), typically at 2e-6.  The smooth, deterministic pins of the backward pass are the op tests
        # and test_every_network_backward_matches_oracle (2e-3); here the bound catches wrong terms and scales.
        grad_tol = [5e-3, 5e-3, 3e-2, 3e-2, 3e-2, 0.25, 0.25]
    else:
        # bf16 storage: op-level parity is pinned in test_ops_gpu.py.  At step level the L1 losses make the
        # gradient discontinuous in the forward activations: bf16 forward noise (~1 %) flips sign(img - fake)
        # on ~0.5 % of the pixels, and as the per-pixel terms do not add coherently at initialisation that
        # alone is a 15-50 % rel-L2 change (cos >= 0.85 measured).  D gradients are (fake - real) differences.
        # (Es: 0.12-0.40 over the fixtures; phase 4 is pure cancellation noise in bf16 at these widths -- 0.6-1.03 --
        # so its bound only catches gross scale errors; the fp32 run of the same fixture pins the logic)
        # (D1/D2: 0.12-0.36 over the ten fixtures)
        grad_tol = [0.5, 0.5, 0.8, 0.5, 0.8, 1.5, 1.5]
    zero_tol = 1e-3 if precision == "fp32" else 5e-2
    # (cosine lower bound, norm-ratio window) per optimizer step of iteration 0
    # measured over the 16 fixtures (bf16): D phases cos 0.931-1.000 / ratio 0.93-1.08, phase 3 cos 0.850-1.000 /
    # ratio 0.96-1.15.  Phase 4 is the small residual of cancelling terms at these widths (the normalisation backward
    # subtracts the mean of an almost uniform gradient; fp32 itself is 3-8 % off): in bf16 neither the norm (ratio
    # 0.15-1.2, every decoder tensor shrinks alike -- per-tensor table under MT_STEP_DIAG=2) nor even the direction is
    # stable: the `nearest` fixture's decoder gradient came out at cos +0.79, +0.79 and -0.90 in three consecutive runs of
    # the same binary (the order of the fp32 atomics in the fused statistics epilogue decides which side of a ReLU a
    # handful of activations land on; with fixed-order reductions two runs are bit-identical -- verified in round 3).
    # So HERE (atomics on, against fp64) phase 4 only gets a gross norm window in bf16; its direction and scale are
    # pinned against the fp32 GPU run in deterministic mode by
    # test_bf16_phase4_gradient_tracks_fp32_run_in_deterministic_mode below, the fp32 run of the same fixture pins the
    # logic to 1e-6, and at real widths bf16 tracks fp32 (tests/test_fullsize_gpu.py).
    dir_tol = [(0.9, (0.85, 1.15))] * 2 + [(0.8, (0.9, 1.25))] * 3 + [(-1.0, (0.1, 3.0))] * 2
    # --dis_sn at these widths: with spectrally normalised weights the adversarial term dominates the generator
    # gradient, and d(logit)/d(image) is piecewise constant in the LeakyReLU pattern of a 4-channel discriminator.
    # Measured on the fp64 oracle: a 1e-7 relative perturbation of the input images moves the phase-3 gradient by
    # 4e-7 or by 9.5e-3 (phase 4: 2e-6 or 0.12) depending on the draw, the reference's own fp32 run sits 1.5e-3 /
    # 8.5e-2 from fp64, ours 1.7e-6 or 1.6e-2 from run to run (atomics order).  The discriminator phases and
    # the losses stay at round-off and keep the strict bounds; the generator phases get a gross-error bound.
    if name == "adain_step_nearest" and precision == "bf16":
        # Re-recorded at --dim 8 in round 4.  Every style-encoder gradient is a linear image of dL/d(mu, logvar) -- 2 x 8 numbers per
        # image -- which at this width is the pixel sum of a near-cancelling field behind the nearest-neighbour up-sampling (each
        # gradient pixel reaches four outputs with identical L1 signs): measured cos 0.848, norm ratio 1.61, rel-L2 0.93 against fp64
        # (full width: cos 0.993, ratio 0.95-1.01, tests/test_fullwidth_step_gpu.py; fp32 run of this fixture: 6e-6).
        grad_tol = list(grad_tol)
        grad_tol[3] = 1.2
        dir_tol = list(dir_tol)
        dir_tol[3] = (0.7, (0.7, 1.9))
    mask_sensitive = name == "adain_step_sn"
    if mask_sensitive:
        # (bf16: the same mask sensitivity inside the discriminator phases -- 0.33 measured on D2 -- so only gross bounds)
        grad_tol = [5e-3, 5e-3, 1e-1, 1e-1, 1e-1, 0.5, 0.5] if precision == "fp32" else [0.6, 0.6, 1.0, 1.0, 1.0, 1.5, 1.5]
    try:
        for it in range(meta["steps"]):
            src = misc.ReplaySource([z[f"rng/{it}/{i}"] for i in range(meta["rng_counts"][it])])
            misc.set_random_source(src)
            seen = []
            for net, opt in M.optimizer.items():
                if not hasattr(opt, "_orig_step"):
                    opt._orig_step = opt.step

                def hooked(closure=None, _net=net, _opt=opt):
                    torch.cuda.synchronize()
                    seen.append((_net, {k: p.grad.detach().clone().cpu() for k, p in M.model[_net].named_parameters()}))
                    return _opt._orig_step()
                opt.step = hooked
            M.update_lr()
            M.set_inputs(batch)
            M.optimize_parameters(it)
            assert src.i == meta["rng_counts"][it]
            got = M.sync_losses()
            t_loss, t_seen, t_state = truth[it]
            # losses: against the reference's recorded scalars AND the fp64 oracle
            for k, v in meta["losses"][it].items():
                assert abs(got[k] - v) <= loss_tol * max(abs(v), 1e-2), f"{name}/{precision} it{it} loss {k}: {got[k]} vs {v}"
                assert abs(got[k] - t_loss[k]) <= loss_tol * max(abs(v), 1e-2), f"{name}/{precision} it{it} loss {k} vs fp64"
            assert [n for n, _ in seen] == meta["grad_nets"][it] == [n for n, _ in t_seen]
            noise_keys = set()
            for j, ((net, g), (_, tg)) in enumerate(zip(seen, t_seen)):
                worst = (0.0, "")
                net_max = max(t.abs().max().item() for t in tg.values())
                ours_all, ref_all = [], []
                for k, ref in tg.items():
                    v = g[k]
                    if ref.abs().max().item() < 1e-4 * net_max:
                        # e.g. the bias in front of an affine-free InstanceNorm: true gradient 0 -- ours must be
                        # negligible too (its Adam trajectory is round-off driven in the reference as well)
                        assert v.abs().max().item() < zero_tol * net_max, f"it{it} step{j} {net}.{k} should vanish"
                        noise_keys.add((net, k))
                        continue
                    ours_all.append(v.double().flatten())
                    ref_all.append(ref.double().flatten())
                    worst = max(worst, (_rel(v, ref), f"it{it} step{j} {net}.{k}"))
                # whole-network gradient vector of this optimizer step
                ours_v, ref_v = torch.cat(ours_all), torch.cat(ref_all)
                net_rel = _rel(ours_v, ref_v)
                cos = (torch.dot(ours_v, ref_v) / (ours_v.norm() * ref_v.norm() + 1e-300)).item()
                ratio = (ours_v.norm() / (ref_v.norm() + 1e-300)).item()
                if os.environ.get("MT_STEP_DIAG"):
                    print(f"DIAG {name} {precision} it{it} step{j} {net}: rel {net_rel:.3e} cos {cos:.4f} ratio {ratio:.3f}")
                    if os.environ.get("MT_STEP_DIAG") == "2" and j >= int(os.environ.get("MT_STEP_DIAG_FROM", "5")):
                        for k, ref in tg.items():
                            print(f"   DIAG2 {net}.{k}: |ours| {g[k].double().norm().item():.3e} |ref| {ref.norm().item():.3e} "
                                  f"rel {_rel(g[k], ref):.3e}")
                # after the first Adam step (~lr*sign(g) per element) the two trajectories differ by
                # round-off-driven sign flips, so later iterations only get a gross-error bound
                tol = grad_tol[j] if it == 0 else max(grad_tol[j], 0.5)
                if it > 0 and precision == "bf16":
                    continue        # second iteration in bf16: the loss scalars above are the check
                assert net_rel <= tol, f"{name}/{precision} it{it} step{j} {net}: gradient rel-L2 error {net_rel} vs fp64 oracle"
                # rel-L2 alone is vacuous above 1.0 (an all-zero gradient scores exactly 1): direction and scale are
                # bounded separately, so a dropped or mis-scaled loss term cannot hide behind bf16 noise
                if it == 0 and tol >= 0.5:   # (a rel-L2 bound below 0.5 already implies cos >= 0.87, ratio in [0.5, 1.5])
                    cos_min, (r_lo, r_hi) = dir_tol[j]
                    assert cos >= cos_min, f"{name}/{precision} it{it} step{j} {net}: gradient cosine {cos:.3f} < {cos_min}"
                    assert r_lo <= ratio <= r_hi, f"{name}/{precision} it{it} step{j} {net}: gradient norm ratio {ratio:.3f}"
                # per tensor: looser (one LeakyReLU mask flipping on a 2-pixel map moves a bias gradient by 10 %)
                assert worst[0] <= max(40 * tol, 0.5) if precision == "fp32" else True, \
                    f"{name}/{precision}: worst per-tensor gradient error {worst}"
            if precision == "fp32" and it == 0:
                # after ONE Adam step the update is ~lr*sign(g): compare the parameter deltas
                for net in M.model:
                    if "discriminator" not in net:
                        continue        # ~lr*sign(g) updates: a 1 % gradient change flips the sign of ~1 % of them,
                                        # and the generator-phase gradients move by that much (see grad_tol above)
                    init = sub(z, f"init/{net}")
                    for k, v in M.model[net].state_dict().items():
                        if (net, k) in noise_keys:
                            continue
                        d_ours = v.detach().cpu().double() - init[k].double()
                        d_ref = t_state[net][k] - init[k].double()
                        # Adam's first update is lr*g/(|g| + 1e-8): elements whose update is visibly shorter than lr have
                        # a round-off-sized gradient (e.g. a class-head bias whose two BCE terms cancel exactly) -- skip them
                        keep = d_ref.abs() > 0.9 * d_ref.abs().max()
                        d_ours, d_ref = d_ours[keep], d_ref[keep]
                        assert _rel(d_ours, d_ref) < 5e-2, f"{name} post-step delta {net}.{k}: {_rel(d_ours, d_ref)}"
    finally:
        misc.set_random_source(None)


@pytest.mark.parametrize("name", ["adain_step_d2", "adain_step_ms"])
def test_training_step_matches_reference_in_deterministic_mode(name, tmp_path, hip_device):
    """The same parity bounds with every reduction in fixed order (MT_DETERMINISTIC: the convolutions' fused statistics
    epilogue -- the last user of float atomics -- is replaced by the separate two-stage statistics pass)."""
    from masterthesis_amd import hip_ops as ops
    ops.set_deterministic(True)
    try:
        test_training_step_matches_reference(name, "fp32", tmp_path, hip_device)
    finally:
        ops.set_deterministic(False)


# bf16 phase 4 (backward_decoder_random) against the fp32 GPU run of the SAME fixture, both with fixed-order reductions
# (ADVICE r2).  Measured over the fixtures in deterministic mode (tools/phase4_bf16_diag.py, round 3):
# two bf16 runs are bit-identical (no race, no uninitialised read: the run-to-run flips seen in round 2 were the order of
# the float atomics of the fused statistics epilogue), and against the fp32 run the phase-4 gradient of the width-8
# fixtures keeps its direction and scale -- content encoder cos 0.70-0.91 / norm ratio 0.90-1.14, decoder cos 0.85-1.00 /
# ratio 0.79-1.03.  The three width-4 fixtures (`--dim 4`: nearest, norms, wgangp) are the exception: their 4-channel
# stacks put the phase-4 gradient at the mercy of single LeakyReLU / L1 sign patterns (tools/phase4_terms_diag.py: in
# `nearest` the d l1_recon_z / d img_random field that comes back through the style encoder has cos -0.28 to the fp32
# one while the forward image agrees to cos 0.9987, and the decoder's gradient is that field's near-cancelling pixel
# sum), deterministic but not a statement about the code; their logic is pinned by the fp32 run of the same fixture.
# Round 4: nearest / norms / wgangp were re-recorded at --dim 8 (no fixture is "width-limited" any more) and now
# take the general bounds -- measured on the re-recorded fixtures: nearest cos 0.71 / 0.96, ratio 0.97 / 0.74; wgangp cos 0.75 /
# 0.80, ratio 1.06 / 1.04; `norms` (LayerNorm in the content encoder, InstanceNorm in the decoder's up-sampling blocks: every
# normalisation of phase 4 subtracts a mean over a near-uniform gradient field) cos 0.63 / 0.65, ratio 0.95 / 0.97: its own cosine
# bound, the norm window is the general one.
PHASE4_COS = {"adain_step_norms": (0.55, 0.55)}
PHASE4_FIXTURES = ["adain_step_d2", "adain_step_d4_b2", "base_step_concat_reparam", "adain_step_lsgan", "adain_step_hinge",
                   "adain_step_ragan", "adain_step_sn", "adain_step_dc", "base_step_concat", "adain_step_dropout",
                   "base_step_concat_dropout", "adain_step_bn", "adain_step_ms", "adain_step_nearest", "adain_step_norms",
                   "adain_step_wgangp"]


def _phase4_gradients(name, tmp_path, precision):
    z, meta, M, misc = _build(name, tmp_path, precision)
    seen = []
    for net, opt in M.optimizer.items():
        orig = opt.step

        def hooked(closure=None, _net=net, _orig=orig):
            torch.cuda.synchronize()
            seen.append((_net, torch.cat([p.grad.detach().double().flatten().cpu() for _, p in M.model[_net].named_parameters()
                                          if p.grad is not None])))
            return _orig()
        opt.step = hooked
    misc.set_random_source(misc.ReplaySource([z[f"rng/0/{i}"] for i in range(meta["rng_counts"][0])]))
    try:
        M.update_lr()
        M.set_inputs(sub(z, "batch"))
        M.optimize_parameters(0)
    finally:
        misc.set_random_source(None)
    torch.cuda.synchronize()
    return seen[-2:]            # the two optimizer steps of phase 4: content encoder, decoder


@pytest.mark.parametrize("name", PHASE4_FIXTURES)
def test_bf16_phase4_gradient_tracks_fp32_run_in_deterministic_mode(name, tmp_path, hip_device):
    from masterthesis_amd import hip_ops as ops
    ops.set_deterministic(True)
    try:
        ref = _phase4_gradients(name, tmp_path / "fp32", "fp32")
        got = _phase4_gradients(name, tmp_path / "bf16", "bf16")
    finally:
        ops.set_deterministic(False)
    assert [n for n, _ in ref] == [n for n, _ in got] == ["content_encoder", "decoder"]
    for (net, a), (_, b) in zip(got, ref):
        cos = (torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)).item()
        ratio = (a.norm() / (b.norm() + 1e-300)).item()
        print(f"MT_DIAG phase4 {name} {net}: bf16 vs fp32 cos {cos:+.3f} ratio {ratio:.3f}")
        assert torch.isfinite(a).all() and a.norm().item() > 0
        cos_min = PHASE4_COS.get(name, (0.6, 0.75))[0 if net == "content_encoder" else 1]
        assert cos >= cos_min, f"{name} {net}: bf16 phase-4 gradient cosine {cos:.3f} to the fp32 run < {cos_min}"
        assert 0.7 <= ratio <= 1.3, f"{name} {net}: bf16 phase-4 gradient norm ratio {ratio:.3f} to the fp32 run"


def test_generator_outputs_within_1e3_of_reference(tmp_path, hip_device):
    """north_star: generator outputs within 1e-3 rel of the CPU reference (fp32 path)."""
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd.models.core import networks as N
    ops.set_compute_dtype(torch.float32)
    z, meta = load_gold("nets_forward")
    x, c, zz = (torch.from_numpy(z[f"AdaINDec/in/{k}"]).to(hip_device) for k in ("x", "c", "z"))
    dec = N.AdaINDecoder(3, dim=32, num_domains=4, latent_dim=8).to(hip_device)
    dec.load_state_dict(sub(z, "AdaINDec/P"))
    with torch.no_grad():
        out = ops.to_nchw_f32(dec(x, zz, c)).cpu()
    ref = torch.from_numpy(z["AdaINDec/out/0"])
    assert ((out - ref).abs().max() / ref.abs().max()).item() < 1e-3
    enc = N.ContentEncoder(3, dim=8).to(hip_device).eval()
    enc.load_state_dict(sub(z, "Ec/P"))
    with torch.no_grad():
        out = ops.to_nchw_f32(enc(torch.from_numpy(z["Ec/in/x"]).to(hip_device))).cpu()
    ref = torch.from_numpy(z["Ec/out/0"])
    assert ((out - ref).abs().max() / ref.abs().max()).item() < 1e-3


NET_CASES = {
    # tag -> (constructor, call) with the arguments oracle/gen_golden.py::nets_case used on the reference classes
    "Ec": (lambda N: N.ContentEncoder(3, dim=8), lambda n, i: n(i["x"])),
    "Es": (lambda N: N.ReparameterizedStyleEncoder(3, output_dim=8, dim=8, num_domains=4, norm_layer=None,
                                                   activation="lrelu"), lambda n, i: n(i["x"], i["c"])),
    "AdaINDec": (lambda N: N.AdaINDecoder(3, dim=32, num_domains=4, latent_dim=8), lambda n, i: n(i["x"], i["z"], i["c"])),
    "D": (lambda N: N.Discriminator(3, dim=8, num_domains=4, image_size=64), lambda n, i: n(i["x"])),
    "MsD": (lambda N: N.MultiScaleDiscriminator(3, dim=2, num_domains=4), lambda n, i: n(i["x"])),
    "Dc": (lambda N: N.ContentDiscriminator(dim=8, num_domains=4), lambda n, i: n(i["x"])),
    "EsPlain": (lambda N: N.StyleEncoder(3, output_dim=8, dim=8, num_domains=4, activation="lrelu"),
                lambda n, i: n(i["x"], i["c"])),
    "DecConcat": (lambda N: N.DecoderConcat(3, dim=32, num_domains=4, latent_dim=8),
                  lambda n, i: n(i["x"], i["z"], i["c"])),
    "DecPlain": (lambda N: N.Decoder(3, dim=32, num_domains=4, latent_dim=8), lambda n, i: n(i["x"], i["z"], i["c"])),
}


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 6e-2)])
@pytest.mark.parametrize("tag", list(NET_CASES))
def test_every_network_forward_matches_reference(tag, precision, tol, hip_device):
    """Each network class of the path (incl. the optional MultiScaleDiscriminator, ContentDiscriminator and BaseModel's
    StyleEncoder / DecoderConcat) with the reference's weights on the reference's inputs: every output within 1e-3 of the
    recorded reference output in fp32 (max error relative to the output's max), bf16 within bf16 rounding."""
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd.models.core import misc, networks as N
    ops.set_compute_dtype(torch.float32 if precision == "fp32" else torch.bfloat16)
    z, meta = load_gold("nets_forward")
    case = [c for c in meta["cases"] if c["tag"] == tag][0]
    make, call = NET_CASES[tag]
    net = make(N)
    net.load_state_dict(sub(z, f"{tag}/P"))
    net = net.to(hip_device).eval()
    inp = {k: v.to(hip_device) for k, v in sub(z, f"{tag}/in").items()}
    misc.set_random_source(misc.ReplaySource([z[f"{tag}/rng/{i}"] for i in range(case["n_rng"])]))
    try:
        with torch.no_grad():
            res = call(net, inp)
    finally:
        misc.set_random_source(None)
    res = res if isinstance(res, (tuple, list)) else (res,)
    flat = []
    for r in res:
        flat += list(r) if isinstance(r, (tuple, list)) else [r]
    assert len(flat) == case["n_out"]
    for i, o in enumerate(flat):
        o = (ops.to_nchw_f32(o) if o.dim() == 4 else o.float()).cpu()
        ref = torch.from_numpy(z[f"{tag}/out/{i}"])
        assert o.shape == ref.shape, (tag, i, o.shape, ref.shape)
        err = ((o - ref).abs().max() / ref.abs().max()).item()
        assert err < tol, f"{tag} output {i} ({precision}): max error {err:.2e} of the output range"


@pytest.mark.parametrize("tag", list(NET_CASES))
def test_every_network_backward_matches_oracle(tag, hip_device):
    """Parameter and input gradients of every network class (fp32) against torch autograd through the CPU oracle's
    restatement (itself pinned forward by the recorded reference outputs of the same fixture), float64, for a random
    cotangent on every output.  This is what covers the backward of the networks no step fixture trains at full width:
    the weight-shared three-scale MultiScaleDiscriminator, and eval-mode variants of the rest."""
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd.models.core import misc, networks as N
    from oracle import nets as onets
    ops.set_compute_dtype(torch.float32)
    z, meta = load_gold("nets_forward")
    case = [c for c in meta["cases"] if c["tag"] == tag][0]
    oracle_fn = {
        "Ec": lambda P, i, r: [onets.content_encoder(P, i["x"], None)],
        "Es": lambda P, i, r: list(onets.style_encoder_reparam(P, i["x"], i["c"], r[0])),
        "AdaINDec": lambda P, i, r: [onets.adain_decoder(P, i["x"], i["z"], i["c"])],
        "D": lambda P, i, r: list(onets.discriminator(P, i["x"])),
        "MsD": lambda P, i, r: [t for pair in onets.multi_scale_discriminator(P, i["x"]) for t in pair],
        "Dc": lambda P, i, r: [onets.content_discriminator(P, i["x"])],
        "EsPlain": lambda P, i, r: [onets.style_encoder_plain(P, i["x"], i["c"])],
        "DecConcat": lambda P, i, r: [onets.decoder_concat(P, i["x"], i["z"], i["c"])],
        "DecPlain": lambda P, i, r: [onets.decoder_plain(P, i["x"], i["z"], i["c"])],
    }[tag]
    rng = [torch.from_numpy(z[f"{tag}/rng/{i}"]) for i in range(case["n_rng"])]
    # ---- float64 truth
    P64 = {k: v.double().requires_grad_(True) for k, v in sub(z, f"{tag}/P").items()}
    in64 = {k: v.double().requires_grad_(k in ("x", "z")) for k, v in sub(z, f"{tag}/in").items()}
    outs64 = oracle_fn(P64, in64, [r.double() for r in rng])
    g = torch.Generator().manual_seed(11)
    cot = [torch.randn(o.shape, generator=g) for o in outs64]
    sum((o * c.double()).sum() for o, c in zip(outs64, cot)).backward()
    # ---- product
    make, call = NET_CASES[tag]
    net = make(N)
    net.load_state_dict(sub(z, f"{tag}/P"))
    net = net.to(hip_device).eval()
    inp = {k: v.to(hip_device).requires_grad_(k in ("x", "z")) for k, v in sub(z, f"{tag}/in").items()}
    misc.set_random_source(misc.ReplaySource([z[f"{tag}/rng/{i}"] for i in range(case["n_rng"])]))
    try:
        res = call(net, inp)
    finally:
        misc.set_random_source(None)
    res = res if isinstance(res, (tuple, list)) else (res,)
    flat = []
    for r in res:
        flat += list(r) if isinstance(r, (tuple, list)) else [r]
    # (the product's activations are NHWC-padded views: feed the cotangent through backward() instead of a dot product)
    torch.autograd.backward(flat, [ops.canon(c.to(hip_device)) if c.dim() == 4 else c.to(hip_device) for c in cot])
    gmax = max(p.grad.abs().max().item() for p in P64.values() if p.grad is not None)
    num = den = 0.0
    for k, p in net.named_parameters():
        ref = P64[k].grad
        if ref is None or ref.abs().max().item() < 1e-6 * gmax:
            continue                    # e.g. the bias in front of an affine-free InstanceNorm (true gradient 0)
        assert p.grad is not None, f"{tag}: no gradient for {k}"
        d = (p.grad.detach().cpu().double() - ref)
        num += (d ** 2).sum().item()
        den += (ref ** 2).sum().item()
        assert d.norm().item() <= 2e-2 * ref.norm().item() + 1e-5 * gmax, \
            f"{tag}.{k}: gradient rel L2 {(d.norm() / ref.norm()).item():.2e}"
    assert (num / den) ** 0.5 < 2e-3, f"{tag}: parameter gradient rel L2 {(num / den) ** 0.5:.2e}"
    for k in ("x", "z"):
        if k in inp and in64[k].grad is not None:
            got = inp[k].grad
            got = (ops.to_nchw_f32(got) if got.dim() == 4 else got.float()).cpu().double()
            rel = ((got - in64[k].grad).norm() / in64[k].grad.norm()).item()
            assert rel < 2e-3, f"{tag}: input gradient d{k} rel L2 {rel:.2e}"
