"""GPU end-to-end: the train.py CLI runs a few iterations on SyntheticDataset, writes the reference's
output tree (checkpoints / logs / images / args.txt), resumes from its own checkpoint, and the inference
surface (forward_random / forward_reference, odd 540x960-like sizes) works."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_cli_checkpoint_resume(tmp_path, hip_device):
    from masterthesis_amd import train
    common = ["--model", "AdaINModel", "--dataset", "SyntheticDataset", "--exp_dir", str(tmp_path), "--name", "run",
              "--batch_size", "2", "--num_domains", "3", "--dim", "8", "--crop_size", "64", "--num_workers", "0",
              "--synthetic_len", "4", "--print_freq", "1", "--save_freq", "2", "--display_freq", "3"]
    train.main(common + ["--n_iters", "3", "--max_iter", "3"])
    run = os.path.join(str(tmp_path), "run")
    ck = os.path.join(run, "checkpoints")
    assert os.path.isfile(os.path.join(run, "args.txt"))
    assert {"model_0.ckpt", "opt_0.ckpt", "model_2.ckpt", "model_4.ckpt"} <= set(os.listdir(ck))
    assert os.listdir(os.path.join(run, "images")) and os.listdir(os.path.join(run, "logs"))
    sd = torch.load(os.path.join(ck, "model_4.ckpt"))
    assert set(sd) == {"content_encoder", "style_encoder", "decoder", "discriminator1", "discriminator2"}
    assert all(torch.isfinite(v).all() for net in sd.values() for v in net.values())
    opt = torch.load(os.path.join(ck, "opt_4.ckpt"))
    st = opt["decoder"]["state"]
    assert all(int(s["step"]) == 8 for s in st.values())        # 4 iterations x 2 decoder steps each
    # resume (weights + optimizer state) and continue two more iterations
    train.main(common + ["--n_iters", "6", "--max_iter", "6", "--resume", os.path.join(ck, "model_4.ckpt"),
                         "--resume_opt", os.path.join(ck, "opt_4.ckpt"), "--last_iter", "4"])
    assert "model_7.ckpt" in os.listdir(ck)


@pytest.mark.parametrize("flags", [["--dis_sn"], ["--gan_mode", "hinge", "--up_type", "nearest"],
                                   ["--gan_mode", "lsgan", "--use_ragan"], ["--ms_dis", "--crop_size", "256"],
                                   ["--use_dis_content", "--crop_size", "224", "--d_iter", "2"], ["--use_dropout"],
                                   ["--enc_norm", "batch", "--dec_norm", "batch", "--dis_norm", "batch", "--batch_size", "2"],
                                   ["--model", "BaseModel", "--concat", "--reparam"], ["--model", "BaseModel"]],
                         ids=["dis_sn", "hinge_nearest", "lsgan_ragan", "ms_dis", "dis_content", "dropout", "batch_norm",
                              "base_concat_reparam", "base_plain"])
def test_train_cli_optional_flags(flags, tmp_path, hip_device):
    """Every optional flag of SURVEY 8f-4 through the command line: three iterations in bf16, a checkpoint, a resume;
    for --dis_sn the checkpoint carries the reference's spectral-norm keys and the power-iteration vectors move."""
    from masterthesis_amd import train
    common = ["--model", "AdaINModel", "--dataset", "SyntheticDataset", "--exp_dir", str(tmp_path), "--name", "run",
              "--batch_size", "1", "--num_domains", "2", "--dim", "8", "--crop_size", "64", "--num_workers", "0",
              "--synthetic_len", "4", "--print_freq", "1", "--save_freq", "2", "--display_freq", "100",
              "--precision", "bf16"] + flags
    train.main(common + ["--n_iters", "3", "--max_iter", "3"])
    ck = os.path.join(str(tmp_path), "run", "checkpoints")
    sd0, sd = torch.load(os.path.join(ck, "model_0.ckpt")), torch.load(os.path.join(ck, "model_2.ckpt"))
    assert all(torch.isfinite(v).all() for net in sd.values() for v in net.values() if v.is_floating_point())
    if "--dis_sn" in flags:
        keys = set(sd["discriminator1"])
        assert {"model.0.block.1.weight_orig", "model.0.block.1.weight_u", "model.0.block.1.weight_v"} <= keys
        assert "model.0.block.1.weight" not in keys
        assert not torch.equal(sd0["discriminator1"]["model.0.block.1.weight_u"],
                               sd["discriminator1"]["model.0.block.1.weight_u"])
    if "--use_dis_content" in flags:
        assert "content_discriminator" in sd
        assert not torch.equal(sd0["content_discriminator"]["model.4.weight"], sd["content_discriminator"]["model.4.weight"])
    train.main(common + ["--n_iters", "5", "--max_iter", "5", "--resume", os.path.join(ck, "model_2.ckpt"),
                         "--resume_opt", os.path.join(ck, "opt_2.ckpt"), "--last_iter", "2"])
    assert "model_4.ckpt" in os.listdir(ck)


def test_sample_cli_end_to_end(tmp_path, hip_device, monkeypatch):
    """sample.py's command line on an image folder (reference sample.py:79-224): a checkpoint written by train.py, two
    images resized to 540x960, translated into two target domains with random styles and with reference images."""
    import numpy as np
    from PIL import Image
    from masterthesis_amd import sample, train
    common = ["--model", "AdaINModel", "--num_domains", "4", "--dim", "8", "--num_workers", "0", "--batch_size", "1"]
    train.main(common + ["--dataset", "SyntheticDataset", "--exp_dir", str(tmp_path), "--name", "run", "--crop_size", "64",
                         "--synthetic_len", "2", "--print_freq", "100", "--save_freq", "1", "--display_freq", "100",
                         "--n_iters", "1", "--max_iter", "1"])
    ckpt = os.path.join(str(tmp_path), "run", "checkpoints", "model_0.ckpt")
    src = tmp_path / "frames"
    os.makedirs(src)
    rng = np.random.default_rng(0)
    for i in range(2):
        Image.fromarray(rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)).save(src / f"f{i}.png")
    out = tmp_path / "out"
    monkeypatch.chdir(tmp_path)           # (the grid modes write ./grid.png like the reference)
    base = common + ["--dataroot", str(src), "--resume", ckpt, "--result_dir", str(out), "--precision", "bf16"]
    sample.Sampler().run(base + ["--targets", "fog", "sun"])
    for trg in ("1", "3"):
        files = sorted(os.listdir(out / "images" / trg))
        assert len(files) == 2, files
        img = Image.open(out / "images" / trg / files[0])
        assert img.size == (960, 540)
    sample.Sampler().run(base + ["--targets", "rain", "--reference", str(src / "f1.png"), "--eval"])
    assert len(os.listdir(out / "images" / "2")) == 2
    sample.Sampler().run(base + ["--gen_grid", "--targets", "cloud", "fog"])
    assert os.path.isfile(tmp_path / "grid.png")


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_inference_surface(precision, tmp_path, hip_device):
    import argparse
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd import models
    a = argparse.Namespace(mode="test", precision=precision, input_dim=3, dim=8, enc_norm="instance", num_domains=4,
                           latent_dim=8, up_type="transpose", dec_norm="layer", use_dropout=False, init_type="normal",
                           init_gain=0.02, resume=None, gpu_ids=[0], batch_size=2, concat=False, reparam=False)
    M = models.AdaINModel(a)
    M.initialize()
    assert list(M.model.keys()) == ["content_encoder", "style_encoder", "decoder"] and not M.optimizer
    img = torch.rand(2, 3, 54, 96, device=hip_device) * 2 - 1        # 540x960 / 10: odd sizes after two stride-2 convs
    c = torch.eye(4, device=hip_device)[[1, 2]]
    with torch.no_grad():
        out, secs, gib = M.forward_random(img, M.get_z_random(2, 8), c)
        out2, _, _ = M.forward_reference(img, img.flip(0), c)
    for o in (out, out2):
        o = ops.to_nchw_f32(o)
        assert o.shape == (2, 3, 54 // 4 * 4 + (4 if 54 % 4 else 0) if False else o.shape[2], o.shape[3])
        assert o.shape[0] == 2 and o.shape[1] == 3 and torch.isfinite(o).all() and o.abs().max() <= 1.0
    assert secs >= 0 and gib >= 0


@pytest.mark.parametrize("comm", ["torch", "native"])
def test_rccl_exchange_path_single_rank(comm, tmp_path, hip_device, monkeypatch):
    """World-size-1 RCCL process group with the gradient exchange forced on: the side-stream all-reduce,
    event waits and per-network Adam ordering run for real; losses must equal a run without the exchange.
    comm = "native": the collective goes through the library's own communicator (mt_comm_* over RCCL).
    Also pins WHERE the exchanges are enqueued: the two buckets of a discriminator from inside its backward pass
    (the 9.4 M-parameter tail first), everything of a phase before the next phase's first launch."""
    monkeypatch.setenv("MT_COMM", comm)
    monkeypatch.setenv("MT_BUCKET_MIN_ELEMS", "1024")       # (the fixture's discriminators are tiny)
    import socket
    import torch.distributed as dist
    from helpers import load_gold, product_args, sub
    from masterthesis_amd import models
    from masterthesis_amd.models.core import misc

    def run(force):
        monkeypatch.setenv("MT_FORCE_REDUCER", "1" if force else "0")
        z, meta = load_gold("adain_step_d2")
        M = models.AdaINModel(product_args(meta["args"], str(tmp_path), "fp32"))
        assert M.reducer.enabled == force
        assert (M.reducer.native is not None) == (force and comm == "native")
        M.initialize()
        log = M.reducer.log = [] if force else None
        for net in M.model:
            M.model[net].load_state_dict(sub(z, f"init/{net}"))
        out = []
        for it in range(2):
            misc.set_random_source(misc.ReplaySource([z[f"rng/{it}/{i}"] for i in range(meta["rng_counts"][it])]))
            M.update_lr()
            M.set_inputs(sub(z, "batch"))
            M.optimize_parameters(it)
            out.append(dict(M.sync_losses()))
        misc.set_random_source(None)
        if force:
            n_d = sum(p.numel() + (-p.numel() % 4) for p in M.model.discriminator1.parameters())
            per_step = len(log) // 2
            for it in range(2):
                ev = log[it * per_step:(it + 1) * per_step]
                kinds = [e[0] if e[0] != "phase" else e[1] for e in ev]
                # discriminator phases: two bucket exchanges each, issued before the next phase is entered
                i1, i2, i3 = kinds.index("discriminator1"), kinds.index("discriminator2"), kinds.index("phase3")
                i4 = kinds.index("phase4")
                r1 = [e[1] for e in ev[i1:i2] if e[0] == "reduce"]
                r2 = [e[1] for e in ev[i2:i3] if e[0] == "reduce"]
                assert len(r1) == 2 and len(r2) == 2, (r1, r2)
                assert sum(x[0] for x in r1) == n_d and r1[0][0] > r1[1][0], r1        # the big tail bucket goes first
                # ... and BOTH leave from inside the backward pass (the second with the first layer's weight gradient)
                for nm, lo in (("discriminator1", i1), ("discriminator2", i2)):
                    done = kinds.index("backward done " + nm)
                    assert kinds[lo:done].count("reduce") == 2, kinds[lo:done + 1]
                # discriminator1 is stepped (1 wait per bucket) before phase 3; discriminator2's waits come after the
                # phase-3 exchange was enqueued (its step is deferred to the start of phase 4)
                assert kinds[i2:i3].count("wait") == 2
                sizes = {n: M.optimizer[n].flat_grad().numel() for n in ("content_encoder", "style_encoder", "decoder")}
                p3 = ev[i3:i4]
                assert sorted(e[1][0] for e in p3 if e[0] == "reduce") == sorted(sizes.values())
                assert [k for k in kinds[i3:i4] if k == "wait"] == ["wait"] * (3 + 2)
                # ... and the decoder's buffer leaves from INSIDE the backward pass: before the phase's exchange point
                ix = kinds.index("exchange content_encoder+style_encoder+decoder")
                early = [e[1][0] for e in ev[i3:ix] if e[0] == "reduce"]
                assert sizes["decoder"] in early, early
                assert sorted(e[1][0] for e in ev[i4:] if e[0] == "reduce") == sorted([sizes["content_encoder"], sizes["decoder"]])
        if M.reducer.native is not None:
            M.reducer.native.close()
        return out

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        a, b = run(True), run(False)
    finally:
        dist.destroy_process_group()
    for la, lb in zip(a, b):
        for k in la:
            assert abs(la[k] - lb[k]) <= 1e-5 * max(1.0, abs(lb[k])), (k, la[k], lb[k])


def test_train_cli_hip_graph_with_checkpoints(tmp_path, hip_device):
    """--hip_graph through the command line: 9 iterations (3 eager, 1 capturing, 5 replayed) with logging / saving /
    image dumps in between; the optimizer checkpoint written after replays carries the right Adam step count, and a run
    resumed from it continues."""
    from masterthesis_amd import train
    from masterthesis_amd.models.core import misc
    misc.set_random_source(None)
    common = ["--model", "AdaINModel", "--dataset", "SyntheticDataset", "--exp_dir", str(tmp_path), "--name", "run",
              "--batch_size", "2", "--num_domains", "2", "--dim", "8", "--crop_size", "64", "--num_workers", "0",
              "--synthetic_len", "4", "--print_freq", "2", "--save_freq", "4", "--display_freq", "3", "--precision", "bf16",
              "--hip_graph"]
    train.main(common + ["--n_iters", "8", "--max_iter", "8"])
    ck = os.path.join(str(tmp_path), "run", "checkpoints")
    assert {"model_0.ckpt", "model_4.ckpt", "model_8.ckpt", "model_9.ckpt"} <= set(os.listdir(ck))
    opt = torch.load(os.path.join(ck, "opt_9.ckpt"))
    assert all(int(s["step"]) == 18 for s in opt["decoder"]["state"].values())      # 9 iterations x 2 decoder steps
    assert all(int(s["step"]) == 9 for s in opt["discriminator2"]["state"].values())
    sd = torch.load(os.path.join(ck, "model_9.ckpt"))
    assert all(torch.isfinite(v).all() for net in sd.values() for v in net.values())
    train.main(common + ["--n_iters", "12", "--max_iter", "12", "--resume", os.path.join(ck, "model_9.ckpt"),
                         "--resume_opt", os.path.join(ck, "opt_9.ckpt"), "--last_iter", "9"])
    assert "model_13.ckpt" in os.listdir(ck)
