"""Pins the CPU oracle (oracle/nets.py, oracle/step.py) against the golden vectors recorded from
the imported reference (oracle/gen_golden.py -> tests/golden/*.npz).  Runs on CPU."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nets, step

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def sub(z, prefix):
    prefix = prefix.rstrip("/") + "/"
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def checksum(t):
    t = torch.as_tensor(t).double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def close(a, b, rtol=1e-4, atol=1e-5, what=""):
    a, b = torch.as_tensor(a).float(), torch.as_tensor(b).float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= atol + rtol * ref, f"{what}: max err {err:.3e} (ref max {ref:.3e})"


NET_FNS = {
    "Ec": lambda P, i, r: [nets.content_encoder(P, i["x"], None)],
    "Es": lambda P, i, r: list(nets.style_encoder_reparam(P, i["x"], i["c"], r[0])),
    "AdaINDec": lambda P, i, r: [nets.adain_decoder(P, i["x"], i["z"], i["c"])],
    "D": lambda P, i, r: list(nets.discriminator(P, i["x"])),
    "MsD": lambda P, i, r: [t for pair in nets.multi_scale_discriminator(P, i["x"]) for t in pair],
    "Dc": lambda P, i, r: [nets.content_discriminator(P, i["x"])],
    "EsPlain": lambda P, i, r: [nets.style_encoder_plain(P, i["x"], i["c"])],
    "DecConcat": lambda P, i, r: [nets.decoder_concat(P, i["x"], i["z"], i["c"])],
    "DecPlain": lambda P, i, r: [nets.decoder_plain(P, i["x"], i["z"], i["c"])],
}


@pytest.mark.parametrize("tag", list(NET_FNS))
def test_network_forward_matches_reference(tag):
    z, meta = load("nets_forward")
    case = [c for c in meta["cases"] if c["tag"] == tag][0]
    P, inp = sub(z, f"{tag}/P"), sub(z, f"{tag}/in")
    rng = [torch.from_numpy(z[f"{tag}/rng/{i}"]) for i in range(case["n_rng"])]
    with torch.no_grad():
        outs = NET_FNS[tag](P, inp, rng)
    assert len(outs) == case["n_out"]
    for i, o in enumerate(outs):
        # relative to the output's own range (no absolute slack: N(0, 0.02)-initialised stacks without normalisation
        # produce outputs of 1e-4 .. 1e-8, where an absolute tolerance would pass anything)
        close(o, z[f"{tag}/out/{i}"], rtol=1e-4, atol=0.0, what=f"{tag} out{i}")


def _run_step_case(name):
    z, meta = load(name)
    a = meta["args"]
    args = step.default_args(**{k: a[k] for k in vars(step.default_args()) if k in a})
    args.model = meta["model"]
    nets_present = sorted({k.split("/")[1] for k in z.files if k.startswith("init/")})
    params = {n: sub(z, f"init/{n}") for n in nets_present}
    M = step.OracleModel(params, args)
    batch = sub(z, "batch")
    return z, meta, M, batch


@pytest.mark.parametrize("name", ["adain_step_d2", "adain_step_d4_b2", "base_step_concat_reparam", "adain_step_lsgan",
                                  "adain_step_hinge", "adain_step_ragan", "adain_step_nearest", "adain_step_sn", "adain_step_dc", "base_step_concat", "adain_step_dropout",
                                  "base_step_concat_dropout", "adain_step_norms", "adain_step_bn", "adain_step_ms", "adain_step_wgangp"])
def test_training_step_matches_reference(name):
    torch.set_num_threads(4)
    z, meta, M, batch = _run_step_case(name)
    for it in range(meta["steps"]):
        rng = step.ReplayRng([z[f"rng/{it}/{i}"] for i in range(meta["rng_counts"][it])])
        # capture the gradients consumed by each optimizer step, in call order
        seen = []
        for net, opt in M.opt.items():
            if not hasattr(opt, "_orig_step"):
                opt._orig_step = opt.step

            def hooked(_net=net, _opt=opt):
                seen.append((_net, {k: (p.grad.detach().clone() if p.grad is not None else None)
                                    for k, p in M.P[_net].items()}))
                return _opt._orig_step()
            opt.step = hooked
        M.update_lr()
        M.set_inputs(batch)
        M.optimize_parameters(it, rng)
        assert rng.i == meta["rng_counts"][it], "random draws consumed in a different number than the reference"
        for k, v in meta["losses"][it].items():
            assert abs(M.loss[k] - v) <= 1e-5 + 2e-4 * abs(v), f"{name} it{it} loss {k}: {M.loss[k]} vs {v}"
        assert [n for n, _ in seen] == meta["grad_nets"][it]
        for j, (net, g) in enumerate(seen):
            # absolute slack relative to the largest gradient of THIS network (discriminator gradients of the
            # N(0, 0.02)-initialised fixtures are 1e-6 .. 1e-4: a fixed atol would not test them)
            net_max = max([float(np.abs(z[f"grad/{it}/{j}/{net}/{k}"]).max()) for k in g
                           if f"grad/{it}/{j}/{net}/{k}" in z.files] or [0.0])
            for k, v in g.items():
                full, cs = f"grad/{it}/{j}/{net}/{k}", f"gradsum/{it}/{j}/{net}/{k}"
                if full in z.files:
                    close(v, z[full], rtol=2e-3, atol=1e-6 * net_max, what=f"{name} it{it} step{j} {net}.{k} grad")
                elif cs in z.files:
                    np.testing.assert_allclose(checksum(v)[1:], z[cs][1:], rtol=2e-3, err_msg=f"{cs}")
        for net, sd in M.P.items():
            for k, v in sd.items():
                np.testing.assert_allclose(checksum(v.detach())[1:], z[f"aftersum/{it}/{net}/{k}"][1:], rtol=1e-4,
                                           err_msg=f"{name} it{it} post-step {net}.{k}")


def test_torch_rng_reproduces_reference_draws():
    """With the reference's seed the oracle draws the very same tensors in the same order."""
    z, meta, M, batch = _run_step_case("adain_step_d2")
    # the reference seeded with torch.manual_seed(0), built + initialised the nets (consuming draws), then
    # stepped; we cannot replay the init draws, but shapes/order of the step's draws must match
    rec = step.RecordingRng()
    M.update_lr()
    M.set_inputs(batch)
    M.optimize_parameters(0, rec)
    shapes = [tuple(t.shape) for t in rec.log]
    want = [tuple(z[f"rng/0/{i}"].shape) for i in range(meta["rng_counts"][0])]
    assert shapes == want


@pytest.mark.parametrize("which", ["random", "reference"])
def test_sampling_forward_matches_reference(which):
    """forward_random / forward_reference (reference adain_model.py:96-109 as sample.py:79-91 calls them: no .eval(), so the
    content noise is live) recorded from the imported reference on a scaled-down 540 x 960 with odd maps everywhere."""
    z, meta = load("sample_forward")
    a = meta["args"]
    args = step.default_args(**{k: a[k] for k in vars(step.default_args()) if k in a})
    args.model = "AdaINModel"
    M = step.OracleModel({n: sub(z, f"init/{n}") for n in ("content_encoder", "style_encoder", "decoder")}, args)
    inp = sub(z, "in")
    rng = step.ReplayRng([z[f"{which}/rng/{i}"] for i in range(meta[f"{which}_rng"])])
    if which == "random":
        y = M.forward_random(inp["img"], inp["z_r"], inp["c"], rng)
    else:
        y = M.forward_reference(inp["img"], inp["ref"], inp["c"], rng)
    assert rng.i == meta[f"{which}_rng"] == (1 if which == "random" else 2)
    close(y, z[f"{which}/out"], rtol=1e-4, atol=0.0, what=f"forward_{which}")
