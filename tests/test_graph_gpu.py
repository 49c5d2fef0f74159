"""hipGraph execution of the training step (--hip_graph): the captured graph must do what the eager step does --
same loss trajectory from the same seed, fresh noise on every replay, Adam's step count / learning rate carried on
the device, checkpoints still consistent."""
import argparse
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(tmp, hip_graph, **kw):
    a = dict(mode="train", precision="fp32", logdir=tmp, checkpoint_dir=tmp, display_dir=tmp, input_dim=3, output_dim=3,
             dim=8, init_type="normal", init_gain=0.02, num_domains=2, latent_dim=8, up_type="transpose",
             dec_norm="layer", enc_norm="instance", use_dropout=False, batch_size=2, crop_size=64, resume=None,
             resume_opt=None, gpu_ids=[0], dis_norm=None, dis_sn=False, ms_dis=False, num_scales=3,
             use_dis_content=False, lr=1e-4, wd=1e-4, beta1=0.5, beta2=0.999, lr_policy="step", n_iters=100,
             n_iter_decay=600000, last_iter=-1, d_iter=3, lambda_rec=10.0, lambda_cls=1.0, lambda_cls_G=5.0,
             gan_mode="vanilla", use_ragan=False, vgg_loss=None, concat=False, reparam=False, max_iter=100,
             hip_graph=hip_graph)
    a.update(kw)
    return argparse.Namespace(**a)


def _run(tmp, hip_graph, steps, hip_device, **kw):
    from masterthesis_amd import models
    from masterthesis_amd.dataset import SyntheticDataset
    from masterthesis_amd.models.core import misc
    misc.set_random_source(None)                     # a fresh DeviceRandom: its seed comes from the torch seed below
    args = _args(tmp, hip_graph, **kw)
    torch.manual_seed(1234)
    torch.cuda.manual_seed(1234)
    M = models.AdaINModel(args)
    M.initialize()
    ds = SyntheticDataset(args, length=4, seed=3)
    batches = []
    for b in range(2):
        items = [ds[b * 2 + i] for i in range(2)]
        batches.append({k: torch.stack([it[k] for it in items]).to(hip_device) for k in items[0]})
    hist = []
    for it in range(steps):
        M.update_lr()
        M.set_inputs(batches[it % 2])
        M.optimize_parameters(it)
        hist.append(dict(M.sync_losses()))
    torch.cuda.synchronize()
    return M, hist


def test_graph_replay_tracks_eager_run(tmp_path, hip_device):
    steps = 9                                        # 3 eager + 1 capturing + 5 replayed iterations
    Me, he = _run(str(tmp_path / "e"), False, steps, hip_device)
    Mg, hg = _run(str(tmp_path / "g"), True, steps, hip_device)
    assert any("graph" in st for st in Mg._graphs.values()), "no graph was captured"
    for it in range(steps):
        for k in ("d_total", "total_g", "l1_self_rec", "l1_cc_rec", "l1_recon_z", "kl_zs", "g_cls"):
            a, b = hg[it][k], he[it][k]
            assert abs(a - b) <= 2e-3 * abs(b) + 1e-5, f"iteration {it} loss {k}: graph {a} vs eager {b}"
    # the parameters end up in the same place (same draws, same arithmetic; only the order of the fp32 atomics differs,
    # which Adam's ~lr*sign(g) updates amplify on round-off-sized gradients: two EAGER runs differ by ~1e-2 as well)
    for net in Me.model:
        pe = torch.cat([p.detach().flatten() for p in Me.model[net].parameters()]).double()
        pg = torch.cat([p.detach().flatten() for p in Mg.model[net].parameters()]).double()
        rel = ((pe - pg).norm() / pe.norm()).item()
        assert rel < 5e-2, f"{net}: parameters differ by {rel:.2e} after {steps} steps"
    # Adam's step counters: device record and host mirror agree with the eager run
    for name in Me.optimizer:
        oe, og = Me.optimizer[name], Mg.optimizer[name]
        assert og._step_count_mt == oe._step_count_mt, name
        dev_step = int(og._dev.view(torch.int32)[1].item())
        assert dev_step == og._step_count_mt, (name, dev_step, og._step_count_mt)
        sd = og.state_dict()
        assert all(int(s["step"]) == oe._step_count_mt for s in sd["state"].values())


def test_graph_replay_draws_fresh_noise_and_follows_lr(tmp_path, hip_device):
    from masterthesis_amd.models.core import misc
    M, hist = _run(str(tmp_path), True, 6, hip_device)
    st = misc.random_source().state(hip_device)
    c0 = int(st[1].item())
    assert c0 == 6 * 4, f"4 content-noise draws per step expected, counter = {c0}"
    l0 = hist[-1]["l1_recon_z"]
    from masterthesis_amd.dataset import SyntheticDataset
    ds = SyntheticDataset(M.args, length=4, seed=3)
    items = [ds[i] for i in range(2)]
    batch = {k: torch.stack([it[k] for it in items]).to(hip_device) for k in items[0]}
    # a learning-rate change made by the scheduler on the host must reach the replayed Adam launches
    for opt in M.optimizer.values():
        opt.param_groups[0]["lr"] = 0.0
    before = torch.cat([p.detach().flatten() for p in M.model.decoder.parameters()]).clone()
    M.set_inputs(batch)
    M.optimize_parameters(6)
    M.sync_losses()
    assert int(st[1].item()) == c0 + 4
    after = torch.cat([p.detach().flatten() for p in M.model.decoder.parameters()])
    assert torch.equal(before, after), "lr = 0 on the host, but the replayed step still moved the weights"
    l1 = M.loss["l1_recon_z"]
    M.set_inputs(batch)
    M.optimize_parameters(7)
    M.sync_losses()
    l2 = M.loss["l1_recon_z"]
    # same weights (lr 0), same batch: only the random draws differ between the two replays
    assert l1 != l2 and l1 != l0, "replays reuse the same random draws"


def test_graph_mode_is_refused_for_host_injected_draws(tmp_path, hip_device):
    """Parity tests inject recorded draws from the host (ReplaySource): those steps must run eagerly."""
    from masterthesis_amd import models
    from masterthesis_amd.models.core import misc
    M = models.AdaINModel(_args(str(tmp_path), True))
    misc.set_random_source(misc.ReplaySource([]))
    try:
        assert not M._graph_mode()
    finally:
        misc.set_random_source(None)
    assert M._graph_mode()


def test_deterministic_mode_is_bit_reproducible(tmp_path, hip_device):
    """MT_DETERMINISTIC / hip_ops.set_deterministic(True): every reduction runs in a fixed order (two-stage statistics,
    ticketed loss reductions, slab sums; the atomics of the fused convolution-statistics epilogue are switched off), so
    two runs from the same seed give BIT-identical losses and parameters over 10 steps -- in bf16 as well."""
    from masterthesis_amd import hip_ops as ops
    ops.set_deterministic(True)
    try:
        runs = []
        for r in range(2):
            M, hist = _run(str(tmp_path / str(r)), False, 10, hip_device, precision="bf16" if r < 2 else "fp32")
            flat = torch.cat([p.detach().flatten() for net in M.model for p in M.model[net].parameters()]).clone()
            runs.append((hist, flat))
        (h0, p0), (h1, p1) = runs
        for it, (a, b) in enumerate(zip(h0, h1)):
            assert a == b, f"iteration {it}: losses differ between two identical-seed runs: {a} vs {b}"
        assert torch.equal(p0, p1), "parameters differ between two identical-seed deterministic runs"
    finally:
        ops.set_deterministic(False)


def test_inference_graph_matches_eager(tmp_path, hip_device):
    """forward_random / forward_reference replayed from a hipGraph (--hip_graph, eval mode so no noise is drawn): same
    output as the eager call, for changing inputs."""
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd import models
    outs = {}
    ops.set_deterministic(True)          # (fixed-order statistics: the two runs can be compared bit for bit)
    for graph in (False, True):
        a = _args(str(tmp_path), graph, mode="test", precision="bf16", num_domains=4, batch_size=1)
        torch.manual_seed(5)
        M = models.AdaINModel(a)
        M.initialize()
        for net in M.model:
            M.model[net].eval()
        res = []
        with torch.no_grad():
            for k in range(5):
                g = torch.Generator().manual_seed(100 + k)
                img = (torch.rand(1, 3, 96, 160, generator=g) * 2 - 1).to(hip_device)
                z = torch.randn(1, 8, generator=g).to(hip_device)
                c = torch.eye(4)[[k % 4]].to(hip_device)
                r, _, _ = M.forward_random(img, z, c)
                res.append(ops.to_nchw_f32(r).clone())
                f, _, _ = M.forward_reference(img, img.flip(3), c)
                res.append(ops.to_nchw_f32(f).clone())
        outs[graph] = res
        if graph:
            assert sum("graph" in st for st in M._infer_graphs.values()) == 2
    ops.set_deterministic(False)
    for a, b in zip(outs[True], outs[False]):
        assert a.shape == (1, 3, 96, 160)
        assert torch.equal(a, b), f"graphed inference differs: max {(a - b).abs().max().item():.3e}"


@pytest.mark.parametrize("variant", ["ms_dis", "dis_content", "base_concat_dropout", "lsgan_sn"])
def test_graph_mode_covers_the_optional_paths(variant, tmp_path, hip_device):
    """The captured step with the optional flags that change the launch sequence: multi-scale discriminators, the
    content discriminator's d_iter gating (two different graphs: full step / content-discriminator-only), BaseModel with
    dropout (device-side Bernoulli masks), spectral norm (in-place power iteration inside the graph).  Losses stay finite
    and follow the eager run of the same seed."""
    from masterthesis_amd import models
    from masterthesis_amd.dataset import SyntheticDataset
    from masterthesis_amd.models.core import misc
    kw = {"ms_dis": dict(ms_dis=True, ms_dim=4, crop_size=256, batch_size=1, num_domains=4),
          "dis_content": dict(use_dis_content=True, crop_size=224, batch_size=1),
          "base_concat_dropout": dict(concat=True, reparam=True, use_dropout=True),
          "lsgan_sn": dict(gan_mode="lsgan", dis_sn=True)}[variant]
    model_cls = models.BaseModel if variant.startswith("base") else models.AdaINModel
    steps = 14 if variant == "dis_content" else 8
    hist = {}
    for graph in (False, True):
        misc.set_random_source(None)
        args = _args(str(tmp_path / str(graph)), graph, **kw)
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        M = model_cls(args)
        M.initialize()
        ds = SyntheticDataset(args, length=2, seed=9)
        items = [ds[i % 2] for i in range(args.batch_size)]
        batch = {k: torch.stack([it[k] for it in items]).to(hip_device) for k in items[0]}
        out = []
        for it in range(steps):
            M.update_lr()
            M.set_inputs(batch)
            M.optimize_parameters(it)
            out.append(dict(M.sync_losses()))
        hist[graph] = out
        if graph:
            n_graphs = sum("graph" in st for st in M._graphs.values())
            assert n_graphs == (2 if variant == "dis_content" else 1), M._graphs.keys()
    for it in range(steps):
        for k, v in hist[True][it].items():
            assert v == v and abs(v) < 1e6, f"{variant} iteration {it}: {k} = {v}"
            e = hist[False][it][k]
            assert abs(v - e) <= 2e-2 * abs(e) + 1e-3, f"{variant} iteration {it} loss {k}: graph {v} vs eager {e}"
