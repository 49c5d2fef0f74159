"""sample.py's inference path (reference adain_model.py:96-109 as sample.py:50,79-91 calls it) against the reference:

  * the fixture recorded from the imported reference (tests/golden/sample_forward.npz: forward_random / forward_reference on a
    scaled-down 540 x 960 with odd maps everywhere, content noise LIVE -- sample.py never calls .eval()) with replayed draws;
  * at the deployment size 540 x 960, dim 64 (135 x 240 at the bottleneck, 67 x 120 / 33 x 60 inside the style encoder) against
    the CPU oracle (pinned to that fixture by tests/test_oracle_golden.py) run on this host with the same weights and draws.

fp32: within 1e-3 of the output range (north_star's generator bound); bf16: 6e-2 of the range (the bound of the network-forward
tests in test_step_gpu.py)."""
import argparse

import pytest
import torch

from fullwidth_common import build_params
from helpers import load_gold, sub

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-3, "bf16": 6e-2}


def _model(meta_args, precision, params, dev):
    from masterthesis_amd import models
    a = dict(meta_args)
    a.update(mode="test", precision=precision, resume=None, gpu_ids=[0])
    a.setdefault("init_gain", 0.02)
    M = models.AdaINModel(argparse.Namespace(**a))
    M.initialize()
    for net in M.model:
        M.model[net].load_state_dict(params[net])
    return M


def _run(M, which, inp, draws, dev):
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd.models.core import misc
    src = misc.ReplaySource(draws)
    misc.set_random_source(src)
    try:
        with torch.no_grad():
            if which == "random":
                y, secs, gib = M.forward_random(inp["img"].to(dev), inp["z_r"].to(dev), inp["c"].to(dev))
            else:
                y, secs, gib = M.forward_reference(inp["img"].to(dev), inp["ref"].to(dev), inp["c"].to(dev))
    finally:
        misc.set_random_source(None)
    assert src.i == len(draws), f"forward_{which} consumed {src.i} of {len(draws)} recorded draws"
    assert secs >= 0 and gib >= 0
    return ops.to_nchw_f32(y).cpu()


def _check(y, ref, precision, what):
    ref = torch.as_tensor(ref).float()
    assert y.shape == ref.shape and torch.isfinite(y).all(), what
    rng = (ref.max() - ref.min()).item()
    err = (y - ref).abs().max().item()
    assert err <= TOL[precision] * rng, f"{what} {precision}: max err {err:.3e} = {err / rng:.2e} of the output range {rng:.3f}"


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("which", ["random", "reference"])
def test_sampling_matches_recorded_reference(which, precision, hip_device):
    z, meta = load_gold("sample_forward")
    params = {n: sub(z, f"init/{n}") for n in ("content_encoder", "style_encoder", "decoder")}
    M = _model(meta["args"], precision, params, hip_device)
    draws = [torch.from_numpy(z[f"{which}/rng/{i}"]) for i in range(meta[f"{which}_rng"])]
    y = _run(M, which, sub(z, "in"), draws, hip_device)
    _check(y, z[f"{which}/out"], precision, f"forward_{which} vs the recorded reference")


_FULL = {}


def _full_size_oracle():
    """oracle outputs at 540 x 960, dim 64, 4 domains, batch of 2 (sample.py repeats the reference image over the batch), once"""
    if not _FULL:
        from oracle import step as ostep
        params = {k: v for k, v in build_params(4, 256, ms=False, seed=5).items() if "discriminator" not in k}
        g = torch.Generator().manual_seed(11)
        inp = {"img": torch.rand(2, 3, 540, 960, generator=g) * 2 - 1, "ref": (torch.rand(1, 3, 540, 960, generator=g) * 2 - 1).repeat(2, 1, 1, 1),
               "z_r": torch.randn(2, 8, generator=g), "c": torch.eye(4)[[3, 3]]}
        O = ostep.OracleModel(params, ostep.default_args(model="AdaINModel", dim=64, num_domains=4, batch_size=2, crop_size=540))
        torch.manual_seed(77)
        out, draws = {}, {}
        for which in ("random", "reference"):
            rec = ostep.RecordingRng()
            out[which] = (O.forward_random(inp["img"], inp["z_r"], inp["c"], rec) if which == "random" else
                          O.forward_reference(inp["img"], inp["ref"], inp["c"], rec))
            draws[which] = rec.log
        _FULL.update(params=params, inp=inp, out=out, draws=draws)
    return _FULL


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("which", ["random", "reference"])
def test_sampling_at_540x960_matches_oracle(which, precision, hip_device):
    import os
    torch.set_num_threads(max(torch.get_num_threads(), min(os.cpu_count() or 8, 128)))
    F = _full_size_oracle()
    meta_args = dict(input_dim=3, dim=64, enc_norm="instance", num_domains=4, latent_dim=8, up_type="transpose", dec_norm="layer",
                     use_dropout=False, init_type="normal", batch_size=2, concat=False, reparam=False)
    M = _model(meta_args, precision, F["params"], hip_device)
    y = _run(M, which, F["inp"], F["draws"][which], hip_device)
    assert y.shape == (2, 3, 540, 960)
    _check(y, F["out"][which], precision, f"forward_{which} at 540x960 vs the CPU oracle")
