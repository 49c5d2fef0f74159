"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the nn.Module surface
has the reference's state_dict keys/shapes, flags and checkpoint plumbing behave, and the product
refuses to compute without a HIP device (no silent fallback)."""
import os
import re

import pytest
import torch

from helpers import load_gold, product_args, sub

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from masterthesis_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mt_api.h")).read()
    declared = set(re.findall(r"\b(mt_[a-z0-9_]+)\s*\(", header)) - {"mt_padc"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libmt_hip.so does not export {name}"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert lib.mt_version() >= 1


@pytest.mark.parametrize("name", ["adain_step_d2", "adain_step_d4_b2", "base_step_concat_reparam"])
def test_state_dict_keys_match_reference(name, tmp_path):
    from masterthesis_amd import models
    z, meta = load_gold(name)
    args = product_args(meta["args"], str(tmp_path))
    M = getattr(models, meta["model"])(args)
    for net in M.model:
        ref = sub(z, f"init/{net}")
        own = M.model[net].state_dict()
        assert list(own.keys()) == list(ref.keys()), net
        for k in ref:
            assert tuple(own[k].shape) == tuple(ref[k].shape), (net, k)
        M.model[net].load_state_dict(ref)      # strict


def test_network_keys_of_optional_networks():
    from masterthesis_amd.models.core import networks as N
    z, _ = load_gold("nets_forward")
    for tag, net in [("MsD", N.MultiScaleDiscriminator(3, dim=2, num_domains=4)),
                     ("Dc", N.ContentDiscriminator(dim=8, num_domains=4)),
                     ("EsPlain", N.StyleEncoder(3, output_dim=8, dim=8, num_domains=4, activation="lrelu")),
                     ("DecConcat", N.DecoderConcat(3, dim=32, num_domains=4, latent_dim=8)),
                     ("DecPlain", N.Decoder(3, dim=32, num_domains=4, latent_dim=8))]:
        ref = sub(z, f"{tag}/P")
        assert list(net.state_dict().keys()) == list(ref.keys()), tag
        net.load_state_dict(ref)


def test_ops_refuse_cpu_tensors():
    from masterthesis_amd import hip_ops as ops
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.conv2d(torch.zeros(1, 8, 4, 4), torch.zeros(8, 8, 3, 3))
    with pytest.raises(RuntimeError, match="HIP device only"):
        ops.canon(torch.zeros(1, 3, 4, 4))


def test_train_arguments_defaults_and_tree(tmp_path):
    from masterthesis_amd.arguments import TrainArguments
    from masterthesis_amd import dataset, models
    args = TrainArguments().parse(["--model", "AdaINModel", "--dataset", "SyntheticDataset", "--exp_dir", str(tmp_path),
                                   "--name", "t", "--batch_size", "1", "--num_domains", "4", "--use_dis_content"])
    assert args.model is models.AdaINModel and args.dataset is dataset.SyntheticDataset
    assert (args.lr, args.wd, args.beta1, float(args.beta2)) == (1e-4, 1e-4, 0.5, 0.999)
    assert (args.lambda_rec, args.lambda_cls, args.lambda_cls_G, args.d_iter) == (10, 1.0, 5.0, 3)
    assert (args.crop_size, args.load_size, args.latent_dim, args.dim) == (256, 286, 8, 64)
    assert args.gpu_ids == [0]
    for d in ("checkpoints", "logs", "images"):
        assert os.path.isdir(os.path.join(str(tmp_path), "t", d))
    assert os.path.isfile(os.path.join(str(tmp_path), "t", "args.txt"))


def test_synthetic_dataset_contract():
    import argparse
    from masterthesis_amd.dataset import SyntheticDataset
    ds = SyntheticDataset(argparse.Namespace(crop_size=32, num_domains=4, synthetic_len=5))
    assert len(ds) == 5
    a, b = ds[3], ds[3]
    assert a["x1"].shape == (3, 32, 32) and a["x1"].dtype == torch.float32
    assert torch.equal(a["x1"], b["x1"]) and -1 <= a["x1"].min() and a["x1"].max() <= 1
    assert a["y1"].sum() == 1 and a["y2"].sum() == 1 and not torch.equal(a["y1"], a["y2"])


def test_paired_dataset_contract_on_image_folders(tmp_path):
    """PairedDataset on a <dataroot>/<mode>/<domain>/*.png tree (reference dataset.py:97-180): bicubic resize to
    load_size, crop to crop_size, [-1, 1] range, two different domains per item, ``index % len`` wrap-around on the
    shorter domain, dataset length = the longest domain."""
    import argparse
    import numpy as np
    from PIL import Image
    from masterthesis_amd.dataset import PairedDataset, SingleDataset
    rng = np.random.default_rng(0)
    counts = {"cloudy": 3, "rain": 1, "sunny": 2}
    for dom, n in counts.items():
        os.makedirs(tmp_path / "train" / dom)
        for i in range(n):
            Image.fromarray(rng.integers(0, 256, (40, 52, 3), dtype=np.uint8)).save(tmp_path / "train" / dom / f"{i}.png")
        (tmp_path / "train" / dom / "notes.txt").write_text("not an image")
    args = argparse.Namespace(dataroot=str(tmp_path), mode="train", select_domains=None, num_domains=3, load_size=36,
                              crop_size=32, no_flip=False)
    ds = PairedDataset(args, return_paths=True)
    assert len(ds) == 3 and ds.targets == [0, 1, 2]
    for idx in range(7):
        it = ds[idx]
        assert it["x1"].shape == (3, 32, 32) and it["x2"].dtype == torch.float32
        assert -1.0 <= it["x1"].min() and it["x1"].max() <= 1.0
        assert it["y1"].sum() == 1 and it["y2"].sum() == 1 and not torch.equal(it["y1"], it["y2"])
        for x_path, y in ((it["x1_path"], it["y1"]), (it["x2_path"], it["y2"])):
            dom = sorted(counts)[int(y.argmax())]
            assert os.path.basename(os.path.dirname(x_path)) == dom
            files = ds.dataset[int(y.argmax())]            # (directory order, unsorted like the reference's listdir)
            assert len(files) == counts[dom] and x_path == files[idx % counts[dom]]
    # test mode: deterministic centre crop, no flip -> the same tensor twice
    args_t = argparse.Namespace(**{**vars(args), "mode": "train", "no_flip": True})
    one = SingleDataset(args_t)
    one.args = argparse.Namespace(**{**vars(args_t), "mode": "test"})
    path = one.dataset[1][0]
    assert torch.equal(one.load_image(path), one.load_image(path))
    # select_domains restricts the tree; unknown names are refused like the reference's assert
    sel = PairedDataset(argparse.Namespace(**{**vars(args), "select_domains": ["sunny", "cloudy"], "num_domains": 2}))
    assert len(sel) == 3 and sel.targets == [0, 1]
    with pytest.raises(AssertionError):
        PairedDataset(argparse.Namespace(**{**vars(args), "select_domains": ["sunny", "fog"], "num_domains": 2}))


def test_checkpoint_roundtrip_and_module_prefix(tmp_path):
    from masterthesis_amd import models
    z, meta = load_gold("adain_step_d2")
    args = product_args(meta["args"], str(tmp_path))
    M = models.AdaINModel(args)
    for net in M.model:
        M.model[net].load_state_dict(sub(z, f"init/{net}"))
    M.save(7)
    path = os.path.join(args.checkpoint_dir, "model_7.ckpt")
    ck = torch.load(path)
    assert set(ck) == {"content_encoder", "style_encoder", "decoder", "discriminator1", "discriminator2"}
    # a checkpoint written by the reference under DataParallel carries 'module.' prefixes
    ck_dp = {net: {"module." + k: v for k, v in sd.items()} for net, sd in ck.items()}
    torch.save(ck_dp, path)
    M2 = models.AdaINModel(args)
    M2.load(path)
    for net in M.model:
        for (k, a), (_, b) in zip(M.model[net].state_dict().items(), M2.model[net].state_dict().items()):
            assert torch.equal(a, b), (net, k)
    assert os.path.isfile(os.path.join(args.checkpoint_dir, "opt_7.ckpt"))


def test_attribute_dict_and_loss_keys():
    from masterthesis_amd.utils import AttributeDict
    d = AttributeDict()
    d.a = 1
    d["b"] = 2
    assert d.a == 1 and d.b == 2 and d.missing is None and list(d) == ["a", "b"]


def test_checkpoint_module_prefix_option_writes_dataparallel_keys(tmp_path):
    """--ckpt_module_prefix: the other direction of the interchange -- the reference's default GPU run wraps
    every net in nn.DataParallel (functions.py:98-101) and its strict load_state_dict expects 'module.' keys."""
    from masterthesis_amd import models
    z, meta = load_gold("adain_step_d2")
    args = product_args(meta["args"], str(tmp_path))
    args.ckpt_module_prefix = True
    M = models.AdaINModel(args)
    M.save(3)
    ck = torch.load(os.path.join(args.checkpoint_dir, "model_3.ckpt"))
    for net, sd in ck.items():
        assert all(k.startswith("module.") for k in sd), net
        wrapped = torch.nn.DataParallel(torch.nn.Sequential())       # a DataParallel-style key set: module.<key>
        want = {"module." + k for k in sub(z, f"init/{net}")}
        assert set(sd) == want, net
        del wrapped
    M2 = models.AdaINModel(args)
    M2.load(os.path.join(args.checkpoint_dir, "model_3.ckpt"))      # and it still loads here


def test_fused_adam_keeps_loaded_step_count(tmp_path):
    """load_state_dict -> state_dict before any step must not reset Adam's step counter."""
    from masterthesis_amd.optim import FusedAdam
    ps = [torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 3))]
    ref = torch.optim.Adam(ps, lr=1e-3)
    for _ in range(3):
        for p in ps:
            p.grad = torch.randn_like(p)
        ref.step()
    sd = ref.state_dict()
    opt = FusedAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-3)
    opt.load_state_dict(sd)
    out = opt.state_dict()
    assert all(int(s["step"]) == 3 for s in out["state"].values())
    assert opt._step_count_mt == 3
    with pytest.raises(ValueError):
        FusedAdam([{"params": [ps[0]]}, {"params": [ps[1]], "lr": 1.0}])
