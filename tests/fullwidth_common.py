"""Shared by tests/test_fullwidth_step_gpu.py and tools/oracle_fullwidth_noise.py: the FULL-WIDTH training step (dim 64, the real
2048-channel multi-scale discriminators) on the CPU oracle with recorded / replayed draws.  Test infrastructure (imports oracle/)."""
import argparse

import torch


def build_params(num_domains, res, ms=True, seed=0):
    """state dicts of freshly initialised product networks (CPU; N(0, 0.02) like arguments.py:27-28), keyed like Model.model"""
    from masterthesis_amd.models.core import networks as N
    from masterthesis_amd.models.core.functions import init_weights
    torch.manual_seed(seed)
    nets = {"content_encoder": N.ContentEncoder(3, dim=64),
            "style_encoder": N.ReparameterizedStyleEncoder(3, output_dim=8, dim=64, num_domains=num_domains,
                                                           norm_layer=None, activation="lrelu"),
            "decoder": N.AdaINDecoder(3, dim=256, num_domains=num_domains, latent_dim=8)}
    for k in ("discriminator1", "discriminator2"):
        nets[k] = (N.MultiScaleDiscriminator(3, num_domains=num_domains) if ms else
                   N.Discriminator(3, dim=64, num_domains=num_domains, image_size=res))
    params = {}
    for k, n in nets.items():
        init_weights(n, "normal", 0.02)
        params[k] = {kk: v.detach().clone() for kk, v in n.state_dict().items()}
    return params


def make_batch(num_domains, res, batch_size=1, seed=7):
    from masterthesis_amd.dataset import SyntheticDataset
    a = argparse.Namespace(crop_size=res, num_domains=num_domains, synthetic_len=max(batch_size, 4))
    ds = SyntheticDataset(a, length=max(batch_size, 4), seed=seed)
    items = [ds[i] for i in range(batch_size)]
    return {k: torch.stack([it[k] for it in items]) for k in items[0]}


def run_oracle(params, batch, num_domains, res, ms, dtype, replay, batch_size=1):
    """one iteration of the oracle -> (13 loss scalars, [(network, {key: gradient}) per optimizer step], list of the draws)
    replay: None = draw (and record) with torch's CPU generator; else the list of tensors to replay"""
    from oracle import step as ostep
    oa = ostep.default_args(model="AdaINModel", dim=64, num_domains=num_domains, batch_size=batch_size, crop_size=res, ms_dis=ms)
    O = ostep.OracleModel(params, oa, dtype=dtype)
    rng = ostep.RecordingRng() if replay is None else ostep.ReplayRng(replay, dtype=dtype)
    if replay is None:
        torch.manual_seed(12345)
    seen = []
    for net, opt in O.opt.items():
        orig = opt.step

        def hooked(_net=net, _orig=orig):
            seen.append((_net, {k: p.grad.detach().clone() for k, p in O.P[_net].items() if p.requires_grad}))
            return _orig()
        opt.step = hooked
    O.update_lr()
    O.set_inputs(batch)
    O.optimize_parameters(0, rng)
    return dict(O.loss), seen, (rng.log if replay is None else replay)
