"""ORACLE tooling -- generates tests/golden/*.npz by IMPORTING the reference (read-only at
/root/reference) in this container and running it on the CPU.  No-op when the reference is absent
(GPU box).  Nothing of the reference's source is copied; fixtures hold inputs, recorded random
draws and outputs only.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden

Import recipe: stub modules for tensorboardX / torchvision / cv2
(none carries hot-path arithmetic), `Tensor.get_device` shim so the reference's GaussianNoiseLayer
and GANLoss run on CPU tensors (misc.py:25, loss.py:60,62), args as a Namespace.
"""
import argparse
import contextlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_stubs():
    sys.dont_write_bytecode = True

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Writer:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass
    mod("tensorboardX", SummaryWriter=_Writer)
    tv = mod("torchvision")
    tv.utils = mod("torchvision.utils", save_image=lambda *a, **k: None)
    tv.transforms = mod("torchvision.transforms")
    tv.models = mod("torchvision.models")
    tv.models.vgg = mod("torchvision.models.vgg")
    mod("cv2")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    orig = torch.Tensor.get_device

    def get_device(self):
        return self.device if not self.is_cuda else orig(self)
    torch.Tensor.get_device = get_device


def ref_args(**kw):
    a = dict(mode="train", logdir="/tmp/_mt_golden_logs", input_dim=3, output_dim=3, dim=8, init_type="normal",
             init_gain=0.02, num_domains=2, latent_dim=8, up_type="transpose", dec_norm="layer", enc_norm="instance",
             use_dropout=False, batch_size=1, crop_size=64, resume=None, resume_opt=None, gpu_ids=[], dis_norm=None,
             dis_sn=False, ms_dis=False, num_scales=3, use_dis_content=False, lr=1e-4, wd=1e-4, beta1=0.5,
             beta2=0.999, lr_policy="step", n_iters=1000000, n_iter_decay=600000, last_iter=-1, d_iter=3,
             lambda_rec=10.0, lambda_cls=1.0, lambda_cls_G=5.0, lambda_style=5.0, lambda_perceptual=1.0,
             gan_mode="vanilla", use_ragan=False, vgg_loss=None, checkpoint_dir="/tmp", display_dir="/tmp",
             concat=False, reparam=False, max_iter=1000000)
    a.update(kw)
    return argparse.Namespace(**a)


@contextlib.contextmanager
def record_rng(log):
    """Record every tensor drawn through torch.randn / Tensor.normal_ (the reference's only draws
    inside a step: misc.py:25, networks.py:132, adain_model.py:84)."""
    o_randn, o_normal = torch.randn, torch.Tensor.normal_

    def randn(*a, **k):
        t = o_randn(*a, **k)
        log.append(t.detach().clone().numpy())
        return t

    def normal_(self, *a, **k):
        t = o_normal(self, *a, **k)
        log.append(t.detach().clone().numpy())
        return t
    import torch.nn.functional as F_
    o_dropout = F_.dropout

    def dropout(x, p=0.5, training=True, inplace=False):
        # nn.Dropout.forward calls F.dropout; the mask of the fused torch op cannot be observed, so the recording run
        # draws it explicitly with the same law (keep ~ Bernoulli(1 - p), scaled by 1 / (1 - p)) and logs it
        if not training:
            return x
        m = torch.bernoulli(torch.full_like(x, 1.0 - p))
        log.append(m.detach().clone().numpy())
        return x * m / (1.0 - p)
    torch.randn, torch.Tensor.normal_, F_.dropout = randn, normal_, dropout
    try:
        yield
    finally:
        torch.randn, torch.Tensor.normal_, F_.dropout = o_randn, o_normal, o_dropout


def make_batch(B, D, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x1 = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    x2 = torch.rand(B, 3, H, W, generator=g) * 2 - 1
    d1 = torch.randint(0, D, (B,), generator=g)
    d2 = (d1 + 1 + torch.randint(0, D - 1, (B,), generator=g)) % D
    eye = torch.eye(D)
    return {"x1": x1, "x2": x2, "y1": eye[d1], "y2": eye[d2]}


def _flat(prefix, sd, out):
    for k, v in sd.items():
        out[f"{prefix}/{k}"] = v.detach().cpu().clone().numpy()


def checksum(t):
    """[sum, sum|x|, sum x^2] in float64 -- compact pin for tensors too large to commit."""
    t = torch.as_tensor(t).double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def _flat_sums(prefix, sd, out):
    for k, v in sd.items():
        out[f"{prefix}/{k}"] = checksum(v.detach())


@contextlib.contextmanager
def ms_width(dim):
    """The reference's AdaINModel builds MultiScaleDiscriminator without passing ``dim`` (adain_model.py:33-42), so its
    width is the class default 64 -> 2048 channels (44.7 M parameters each: too large for a committed fixture and for a
    float64 CPU run).  The recording run changes ONLY that constructor default for the duration of the model construction, so the
    step logic being pinned (182-199, 278-285, 352-359) is the reference's own, on a narrow discriminator; the product
    reaches the same width through its optional ``--ms_dim`` flag."""
    from models.core import networks as N
    if dim is None:
        yield
        return
    init = N.MultiScaleDiscriminator.__init__
    orig = init.__defaults__
    assert init.__code__.co_varnames[2] == "dim" and orig[0] == 64
    init.__defaults__ = (dim,) + orig[1:]
    try:
        yield
    finally:
        init.__defaults__ = orig


def step_case(name, model_name, steps=2, seed=0, full_grads=(0,), ms_dim=None, **kw):
    """One or two full optimize_parameters() calls of the imported reference."""
    import models  # the reference package
    args = ref_args(**kw)
    if ms_dim is not None:
        args.ms_dim = ms_dim            # (read by nobody in the reference; recorded in meta for the product's --ms_dim)
    torch.manual_seed(seed)
    with ms_width(ms_dim):
        M = getattr(models, model_name)(args)
    M.initialize()
    # nn.Linear layers keep torch's default init (functions.py:72-94 only touches Conv*)
    out = {}
    for net in M.model:
        _flat(f"init/{net}", M.model[net].state_dict(), out)
    batch = make_batch(args.batch_size, args.num_domains, args.crop_size, args.crop_size, seed + 100)
    for k, v in batch.items():
        out[f"batch/{k}"] = v.numpy()
    # grads as consumed by each optimizer.step() (7 per full step), in call order
    grads = []
    for net, opt in M.optimizer.items():
        o_step = opt.step

        def step(closure=None, _net=net, _o=o_step):
            grads.append((_net, {k: (p.grad.detach().clone() if p.grad is not None else None)
                                 for k, p in M.model[_net].named_parameters()}))
            return _o()
        opt.step = step
    meta = {"name": name, "model": model_name, "args": {k: v for k, v in vars(args).items()
                                                        if isinstance(v, (int, float, str, bool, type(None)))},
            "steps": steps, "losses": [], "rng_counts": [], "grad_nets": []}
    for it in range(steps):
        rng = []
        grads.clear()
        with record_rng(rng):
            M.update_lr()
            M.set_inputs(batch)
            M.optimize_parameters(it)
        for i, t in enumerate(rng):
            out[f"rng/{it}/{i}"] = t
        meta["rng_counts"].append(len(rng))
        meta["losses"].append({k: float(v) for k, v in M.loss.items()})
        nets_called = []
        for j, (net, g) in enumerate(grads):
            nets_called.append(net)
            for k, v in g.items():
                if v is not None:
                    if it in full_grads:
                        out[f"grad/{it}/{j}/{net}/{k}"] = v.numpy()
                    else:
                        out[f"gradsum/{it}/{j}/{net}/{k}"] = checksum(v)
        meta["grad_nets"].append(nets_called)
        for net in M.model:
            _flat_sums(f"aftersum/{it}/{net}", M.model[net].state_dict(), out)
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, losses[0]={meta['losses'][0]}")


def nets_case(name="nets_forward", seed=3):
    """Forward outputs of every network class on the path (eval mode: no noise; eps recorded)."""
    from models.core import networks as N
    from models.core.functions import init_weights
    torch.manual_seed(seed)
    out, meta = {}, {"name": name, "cases": []}
    g = torch.Generator().manual_seed(seed)

    def rnd(*s):
        return torch.randn(*s, generator=g)

    def add(tag, net, inputs, fwd):
        init_weights(net, "normal", 0.02)
        net.eval()
        _flat(f"{tag}/P", net.state_dict(), out)
        for k, v in inputs.items():
            out[f"{tag}/in/{k}"] = v.numpy()
        rng = []
        with torch.no_grad(), record_rng(rng):
            res = fwd(net)
        res = res if isinstance(res, (tuple, list)) else (res,)
        flat = []
        for r in res:
            flat += list(r) if isinstance(r, (tuple, list)) else [r]
        for i, r in enumerate(flat):
            out[f"{tag}/out/{i}"] = r.numpy()
        for i, t in enumerate(rng):
            out[f"{tag}/rng/{i}"] = t
        meta["cases"].append({"tag": tag, "n_out": len(flat), "n_rng": len(rng)})

    x = rnd(2, 3, 64, 64)
    c = torch.eye(4)[[1, 3]]
    z = rnd(2, 8)
    add("Ec", N.ContentEncoder(3, dim=8), {"x": x}, lambda n: n(x))
    add("Es", N.ReparameterizedStyleEncoder(3, output_dim=8, dim=8, num_domains=4, norm_layer=None,
                                            activation="lrelu"), {"x": x, "c": c}, lambda n: n(x, c))
    zc = rnd(2, 32, 16, 16)
    add("AdaINDec", N.AdaINDecoder(3, dim=32, num_domains=4, latent_dim=8), {"x": zc, "z": z, "c": c},
        lambda n: n(zc, z, c))
    add("D", N.Discriminator(3, dim=8, num_domains=4, image_size=64), {"x": x}, lambda n: n(x))
    x256 = rnd(1, 3, 256, 256)
    add("MsD", N.MultiScaleDiscriminator(3, dim=2, num_domains=4), {"x": x256}, lambda n: n(x256))
    zc56 = rnd(2, 8, 56, 56)
    add("Dc", N.ContentDiscriminator(dim=8, num_domains=4), {"x": zc56}, lambda n: n(zc56))
    add("EsPlain", N.StyleEncoder(3, output_dim=8, dim=8, num_domains=4, activation="lrelu"), {"x": x, "c": c},
        lambda n: n(x, c))
    zc2 = rnd(2, 32, 16, 16)
    add("DecConcat", N.DecoderConcat(3, dim=32, num_domains=4, latent_dim=8), {"x": zc2, "z": z, "c": c},
        lambda n: n(zc2, z, c))
    # BaseModel without --concat: forward only (the reference's training step raises there on torch >= 2: in-place
    # `out += residual` on a ReLU output, blocks.py:190,207)
    zc3 = rnd(2, 32, 16, 16)
    add("DecPlain", N.Decoder(3, dim=32, num_domains=4, latent_dim=8), {"x": zc3, "z": z, "c": c},
        lambda n: n(zc3, z, c))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB")


def sample_case(name="sample_forward", seed=21, H=60, W=108, dim=8, num_domains=4, bs=2):
    """``forward_random`` / ``forward_reference`` of the imported reference (adain_model.py:96-109) the way sample.py:79-91 calls
    them: a non-train model (encoders + decoder only), no ``.eval()`` anywhere (sample.py never calls it: the content encoder's
    GaussianNoiseLayer stays active, Appendix D-12) -- the draws are recorded.  H x W is a scaled-down 540 x 960 (9:16 with ODD
    maps after the two stride-2 layers: 15 x 27), so reflection padding / transposed convolutions / the style encoder's pools
    see the odd sizes of the real sampling resolution.  Only ``torch.cuda.memory_reserved`` is patched (it raises without a GPU)."""
    import models
    args = ref_args(mode="test", dim=dim, num_domains=num_domains, batch_size=bs, crop_size=H)
    torch.manual_seed(seed)
    M = models.AdaINModel(args)
    M.initialize()
    out = {}
    for net in M.model:
        _flat(f"init/{net}", M.model[net].state_dict(), out)
    g = torch.Generator().manual_seed(seed + 1)
    img = torch.rand(bs, 3, H, W, generator=g) * 2 - 1
    ref = torch.rand(bs, 3, H, W, generator=g) * 2 - 1
    z_r = torch.randn(bs, args.latent_dim, generator=g)
    c = torch.zeros(bs, num_domains)
    c[:, 2] = 1                                        # sample.py:73-77 load_target: the same target for the whole batch
    out.update({"in/img": img.numpy(), "in/ref": ref.numpy(), "in/z_r": z_r.numpy(), "in/c": c.numpy()})
    meta = {"name": name, "model": "AdaINModel", "args": {k: v for k, v in vars(args).items()
                                                          if isinstance(v, (int, float, str, bool, type(None)))}}
    reserved = torch.cuda.memory_reserved
    torch.cuda.memory_reserved = lambda *a, **k: 0
    try:
        for tag, call in (("random", lambda: M.forward_random(img, z_r, c)), ("reference", lambda: M.forward_reference(img, ref, c))):
            rng = []
            with torch.no_grad(), record_rng(rng):
                y, secs, gib = call()
            out[f"{tag}/out"] = y.numpy()
            for i, t in enumerate(rng):
                out[f"{tag}/rng/{i}"] = t
            meta[f"{tag}_rng"] = len(rng)
    finally:
        torch.cuda.memory_reserved = reserved
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB, rng draws {meta['random_rng']} / {meta['reference_rng']}")


def main():
    if not os.path.isdir(REF):
        print("reference not present -- nothing to generate")
        return 0
    os.makedirs(OUT, exist_ok=True)
    _install_stubs()
    torch.set_num_threads(4)
    only = set(sys.argv[1:])          # optional: names of the fixtures to (re)generate

    def want(name):
        return not only or name in only
    if want("nets_forward"):
        nets_case()
    if want("sample_forward"):
        sample_case()
    if want("adain_step_d2"):
        step_case("adain_step_d2", "AdaINModel", steps=2, seed=0, num_domains=2, batch_size=1)
    # dim 4: channel counts that are not multiples of 8 (exercises the channel padding)
    if want("adain_step_d4_b2"):
        step_case("adain_step_d4_b2", "AdaINModel", steps=1, seed=1, num_domains=4, batch_size=2, dim=4)
    if want("base_step_concat_reparam"):
        step_case("base_step_concat_reparam", "BaseModel", steps=1, seed=2, num_domains=2, batch_size=1, dim=4,
                  concat=True, reparam=True)
    if want("base_step_concat"):
        # BaseModel with the plain StyleEncoder (no --reparam): style codes without KL / reparameterisation
        step_case("base_step_concat", "BaseModel", steps=1, seed=10, num_domains=2, batch_size=1, dim=4, concat=True)
    # optional GAN objectives: --gan_mode lsgan / hinge
    if want("adain_step_lsgan"):
        step_case("adain_step_lsgan", "AdaINModel", steps=1, seed=4, num_domains=2, batch_size=1, dim=4,
                  gan_mode="lsgan")
    if want("adain_step_hinge"):
        step_case("adain_step_hinge", "AdaINModel", steps=1, seed=5, num_domains=2, batch_size=1, dim=4,
                  gan_mode="hinge")
    if want("adain_step_nearest"):
        step_case("adain_step_nearest", "AdaINModel", steps=1, seed=7, num_domains=2, batch_size=1, dim=8,
                  up_type="nearest")
    if want("adain_step_sn"):
        # --dis_sn: two steps so the power-iteration vectors carried across calls and steps are pinned too
        step_case("adain_step_sn", "AdaINModel", steps=2, seed=8, num_domains=2, batch_size=1, dim=4, dis_sn=True)
    if want("adain_step_dc"):
        # --use_dis_content: iteration 0 is a full step with the content-adversarial generator term, iteration 1 only
        # updates the content discriminator (d_iter gating); 224x224 because Dc needs a >= 53x53 content map
        step_case("adain_step_dc", "AdaINModel", steps=2, seed=9, num_domains=2, batch_size=1, dim=4, crop_size=224,
                  use_dis_content=True)
    if want("adain_step_dropout"):
        step_case("adain_step_dropout", "AdaINModel", steps=1, seed=11, num_domains=2, batch_size=1, dim=4,
                  use_dropout=True)
    if want("base_step_concat_dropout"):
        step_case("base_step_concat_dropout", "BaseModel", steps=1, seed=12, num_domains=2, batch_size=1, dim=4,
                  concat=True, reparam=True, use_dropout=True)
    if want("adain_step_norms"):
        # non-default normalisation flags: LayerNorm in the content encoder, InstanceNorm in the decoder's upsampling
        # blocks and in the discriminators
        step_case("adain_step_norms", "AdaINModel", steps=1, seed=13, num_domains=2, batch_size=1, dim=8,
                  enc_norm="layer", dec_norm="instance", dis_norm="instance")
    if want("adain_step_bn"):
        # BatchNorm2d everywhere a norm flag reaches (running statistics land in the post-step checksums); two steps
        step_case("adain_step_bn", "AdaINModel", steps=2, seed=14, num_domains=2, batch_size=2, dim=4,
                  enc_norm="batch", dec_norm="batch", dis_norm="batch")
    if want("adain_step_ragan"):
        step_case("adain_step_ragan", "AdaINModel", steps=1, seed=6, num_domains=2, batch_size=1, dim=4,
                  use_ragan=True)
    if want("adain_step_ms"):
        # --ms_dis (the north-star discriminator): 3-scale loss sums (adain_model.py:182-199), the multi-scale generator
        # term (278-285) and the discriminator1-in-phase-4 quirk (352-353), 4 domains, two iterations at 256x256 (the
        # smallest size the 6-layer k4 s2 stack accepts at 3 scales)
        step_case("adain_step_ms", "AdaINModel", steps=2, seed=15, num_domains=4, batch_size=1, dim=4, crop_size=256,
                  ms_dis=True, ms_dim=4)
    if want("adain_step_wgangp"):
        # --gan_mode wgangp: GANLoss returns -mean / +mean (loss.py:53-57); the reference has no gradient penalty
        step_case("adain_step_wgangp", "AdaINModel", steps=1, seed=16, num_domains=2, batch_size=1, dim=8,
                  gan_mode="wgangp")
    return 0


if __name__ == "__main__":
    sys.exit(main())
