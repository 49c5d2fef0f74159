"""The FULL-WIDTH training step against the CPU oracle.  The reference fixtures of test_step_gpu.py are
`--dim 8` / 64x64 (`--ms_dim 2..4` for the multi-scale discriminators): they pin the step LOGIC, but the kernels the bench runs
-- 256x256 ping-pong tiles, grouped / parked weight gradients, one slab sum per weight, the merged mini-image batches of the
multi-scale discriminators, the thin-head streaming kernels, split-K -- are selected by shapes those fixtures never produce.  Here
the product runs ONE step at dim 64, 256x256, 4 domains, `--ms_dis` with the real 64 -> 2048 channel discriminators
(reference adain_model.py:136-394, networks.py:445-466) and is compared with the oracle (oracle/step.py, pinned to the reference by
tests/golden) run on this host's CPU with the SAME initial weights and the SAME replayed draws:

  * batch_size 1 (2 images per network call, ~30 s of oracle time on the GPU box's host): fp32, bf16, and fp32 / bf16 once more with
    every grouping optimisation switched off (MT_WGRAD_GROUP=0, MT_MSD_MERGE=0, MT_NORM_ONEPASS=0) -- the fallback paths;
  * batch_size 4 (the decoder batches of phase 3 are 16 images = one full round of 256x256 tiles: the bench's K1 shape with its
    data gradient, grouped weight gradients and the one-pass norm backward inside a step): fp32 and bf16.

Bounds.  The oracle runs in fp32 here (fp64 takes minutes); tools/oracle_fullwidth_noise.py measured the fp32 oracle against the
fp64 one at this size (see FP32_ORACLE_NOISE below), which is the floor of the comparison; the fp32 product must stay within a
small multiple of it.  bf16: cosine / norm-ratio windows per optimizer step."""
import os

import pytest
import torch

from fullwidth_common import build_params, make_batch, run_oracle

pytestmark = pytest.mark.gpu

RES, DOMAINS = 256, 4
# tools/oracle_fullwidth_noise.py (fp32 oracle vs fp64 oracle, batch_size 1, this size), rel-L2 per optimizer step:
# D1, D2 | Ec, Es, Dec (phase 3) | Ec, Dec (phase 4); all 13 loss scalars agree to < 2e-7.  The phase-4 content-encoder gradient is
# cancellation-dominated at FULL width too (cos 0.9951 between the two oracles): the 0.25 bound of the narrow fixtures was not a
# width artefact, and no fp32 implementation can be pinned closer than this to another fp32 one there.
FP32_ORACLE_NOISE = [1.3e-5, 1.7e-5, 3.9e-3, 5.7e-4, 1.0e-3, 9.9e-2, 9.5e-3]
# fp32 product vs fp32 oracle: rel-L2 per optimizer step
# measured (round 4, MI355X): b1 2.1e-4 / 7.0e-5 | 1.2e-2 / 2.1e-3 / 6.9e-3 | 0.128 / 1.3e-2; b4 2.7e-4 / 3.4e-4 | 8.5e-3 / 2.6e-4 / 1.4e-3 |
# 0.119 / 6.1e-3; with the fallback paths 2.8e-5 / 5.8e-5 | 6.3e-3 / 1.8e-3 / 1.6e-3 | 0.110 / 1.0e-2 -- i.e. 1.2-3x the fp32 oracle's own
# distance from fp64 in the generator phases
FP32_TOL = [2e-3, 2e-3, 3e-2, 1e-2, 2e-2, 0.3, 4e-2]
# bf16 product vs fp32 oracle: (cosine lower bound, norm-ratio window) per optimizer step
# measured: D phases cos 0.9992-0.9997; phase 3 content encoder cos 0.895-0.900, ratio 0.997-0.999 (the L1 terms and ReLU masks make
# the gradient piecewise constant in the bf16 forward activations: tests/test_step_gpu.py)
BF16_DIR = [(0.99, (0.95, 1.05))] * 2 + [(0.8, (0.85, 1.25))] * 3 + [(0.5, (0.6, 1.5))] * 2
STEP_NETS = ["discriminator1", "discriminator2", "content_encoder", "style_encoder", "decoder", "content_encoder", "decoder"]

_ORACLE = {}


def _oracle(batch_size):
    """fp32 oracle step at this batch size, once per test session"""
    if batch_size not in _ORACLE:
        params = build_params(DOMAINS, RES, ms=True)
        batch = make_batch(DOMAINS, RES, batch_size)
        losses, grads, draws = run_oracle(params, batch, DOMAINS, RES, True, torch.float32, None, batch_size)
        assert [n for n, _ in grads] == STEP_NETS
        _ORACLE[batch_size] = (params, batch, losses, grads, draws)
    return _ORACLE[batch_size]


def _product_step(batch_size, precision, tmp_path, dev, fallbacks):
    import argparse
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from masterthesis_amd import hip_ops as ops, models
    from masterthesis_amd.models.core import misc, networks
    params, batch, _, _, draws = _oracle(batch_size)
    o = argparse.Namespace(precision=precision, num_domains=DOMAINS, batch_size=batch_size, crop_size=RES, ms_dis=True)
    args = bench.model_args(o, str(tmp_path))
    torch.manual_seed(0)
    M = models.AdaINModel(args)
    M.initialize()
    for net in M.model:
        M.model[net].load_state_dict(params[net])
    seen = []
    for net, opt in M.optimizer.items():
        orig = opt.step

        def hooked(closure=None, _net=net, _orig=orig):
            torch.cuda.synchronize()
            seen.append((_net, {k: p.grad.detach().clone().cpu() for k, p in M.model[_net].named_parameters()}))
            return _orig()
        opt.step = hooked
    saved = (ops._WGRAD_GROUP_ON[0], networks._MSD_MERGE[0], ops._NORM_ONEPASS_ON[0])
    if fallbacks:
        ops.set_wgrad_group(False)
        networks._MSD_MERGE[0] = False
        ops.set_norm_onepass(False)
    src = misc.ReplaySource(draws)
    misc.set_random_source(src)
    try:
        ops.hbm_timer_start()
        ops.oplog_start()
        M.update_lr()
        M.set_inputs({k: v.to(dev) for k, v in batch.items()})
        M.optimize_parameters(0)
        got = dict(M.sync_losses())
        log = ops.oplog_stop()
        used = ops.hbm_timer_stop()
    finally:
        misc.set_random_source(None)
        ops.set_wgrad_group(saved[0])
        networks._MSD_MERGE[0] = saved[1]
        ops.set_norm_onepass(saved[2])
    assert src.i == len(draws), "the product consumed a different number of random draws than the oracle"
    return got, seen, used, log


def _vec(grads, ref):
    return torch.cat([grads[k].double().flatten() for k in ref])


CASES = [(1, "fp32", False), (1, "bf16", False), (1, "fp32", True), (1, "bf16", True), (4, "fp32", False), (4, "bf16", False)]


@pytest.mark.parametrize("batch_size,precision,fallbacks", CASES,
                         ids=[f"b{b}_{p}{'_fallbacks' if f else ''}" for b, p, f in CASES])
def test_fullwidth_step_matches_oracle(batch_size, precision, fallbacks, tmp_path, hip_device):
    torch.set_num_threads(max(torch.get_num_threads(), min(os.cpu_count() or 8, 128)))
    _, _, t_loss, t_grads, _ = _oracle(batch_size)
    got, seen, used, log = _product_step(batch_size, precision, tmp_path, hip_device, fallbacks)
    loss_tol = 1e-3 if precision == "fp32" else 3e-2
    assert set(t_loss) <= set(got)
    for k, v in t_loss.items():
        assert abs(got[k] - v) <= loss_tol * max(abs(v), 1e-2), f"b{batch_size}/{precision} loss {k}: {got[k]} vs oracle {v}"
    assert [n for n, _ in seen] == STEP_NETS
    diag = os.environ.get("MT_STEP_DIAG")
    bad = []
    for j, ((net, g), (_, tg)) in enumerate(zip(seen, t_grads)):
        assert set(tg) <= set(g), (net, set(tg) - set(g))
        ours, ref = _vec(g, tg), _vec(tg, tg)
        assert torch.isfinite(ours).all(), f"step {j} {net}: non-finite gradient"
        rel = ((ours - ref).norm() / ref.norm()).item()
        cos = (torch.dot(ours, ref) / (ours.norm() * ref.norm())).item()
        ratio = (ours.norm() / ref.norm()).item()
        if diag:
            print(f"DIAG fullwidth b{batch_size} {precision}{' fallbacks' if fallbacks else ''} step{j} {net}: rel {rel:.3e} "
                  f"cos {cos:.5f} ratio {ratio:.4f}")
        if precision == "fp32":
            if rel > FP32_TOL[j]:
                bad.append(f"b{batch_size}/fp32 step {j} {net}: gradient rel-L2 {rel:.3e} vs the fp32 oracle (bound {FP32_TOL[j]})")
        else:
            cmin, (lo, hi) = BF16_DIR[j]
            if not (cos >= cmin and lo <= ratio <= hi):
                bad.append(f"b{batch_size}/bf16 step {j} {net}: cos {cos:.4f} ratio {ratio:.4f}")
    assert not bad, "; ".join(bad)
    # the run really went through the paths this test exists for (or, with fallbacks, around them)
    kinds = {}
    for kind, d, _ms in log:
        kinds.setdefault(kind, []).append(d)
    grouped = [d for d in kinds.get("wgrad", []) if len(d) > 16 and d[-1] >= 2]
    merged = [d for d in kinds.get("fwd", []) if d[7] == 4 and d[9] == 4]          # kh = 4, stride = 4: the mini-image batches
    if fallbacks:
        assert not grouped and not merged and "norm_bwd_onepass" not in used
        assert "wgrad_sum" not in kinds
    else:
        assert merged, "the multi-scale discriminators did not take the merged mini-image path"
        assert "wgrad_sum" in kinds, "no weight shared one slab sum across its uses"
        if batch_size >= 4 and precision == "bf16":      # (the 256x256 weight-gradient kernel and the one-pass backward are bf16 kernels)
            assert grouped, "no grouped weight-gradient launch at batch_size 4"
            assert "norm_bwd_onepass" in used
