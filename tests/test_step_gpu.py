"""GPU parity of the full G+D training step (through the C ABI) against the golden vectors recorded
from the reference: loss scalars, the gradients consumed by each of the 7 Adam steps, and the
post-step parameters.  fp32 path: the north-star tolerance (1e-3 relative); bf16 path: bf16 rounding."""
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


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("name", ["adain_step_d2", "adain_step_d4_b2", "base_step_concat_reparam"])
def test_training_step_matches_reference(name, precision, tmp_path, hip_device):
    z, meta, M, misc = _build(name, tmp_path, precision)
    batch = sub(z, "batch")
    loss_tol = 1e-3 if precision == "fp32" else 3e-2
    grad_tol = 2e-3 if precision == "fp32" else 8e-2
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
            for k, v in meta["losses"][it].items():
                assert abs(got[k] - v) <= loss_tol * max(abs(v), 1e-2), f"{name}/{precision} it{it} loss {k}: {got[k]} vs {v}"
            assert [n for n, _ in seen] == meta["grad_nets"][it]
            worst = (0.0, "")
            for j, (net, g) in enumerate(seen):
                for k, v in g.items():
                    full = f"grad/{it}/{j}/{net}/{k}"
                    if full in z.files:
                        ref = torch.from_numpy(z[full])
                        if ref.abs().max() < 1e-7:      # bias before an affine-free InstanceNorm: exact zero gradient
                            assert v.abs().max() < 1e-4, f"{full} should vanish"
                            continue
                        r = _rel(v, ref)
                        worst = max(worst, (r, full))
            assert worst[0] <= grad_tol, f"{name}/{precision} it{it}: worst gradient rel-L2 error {worst}"
            if precision == "fp32":
                for net in M.model:
                    for k, v in M.model[net].state_dict().items():
                        np.testing.assert_allclose(checksum(v)[1:], z[f"aftersum/{it}/{net}/{k}"][1:], rtol=2e-4,
                                                   err_msg=f"{name} it{it} post-step {net}.{k}")
    finally:
        misc.set_random_source(None)


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
