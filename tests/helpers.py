"""Shared helpers for the golden-fixture tests (pure data handling; no reference code)."""
import argparse
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_gold(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def sub(z, prefix):
    prefix = prefix.rstrip("/") + "/"
    return {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}


def checksum(t):
    t = torch.as_tensor(t).double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def product_args(meta_args, tmpdir, precision="fp32", mode="train"):
    """Namespace for masterthesis_amd models from the args recorded in a fixture."""
    a = dict(meta_args)
    a.setdefault("dis_norm", None)
    a.update(mode=mode, precision=precision, logdir=os.path.join(tmpdir, "logs"),
             checkpoint_dir=os.path.join(tmpdir, "ckpt"), display_dir=os.path.join(tmpdir, "img"),
             gpu_ids=[0], resume=None, resume_opt=None, vgg_loss=None)
    for d in (a["logdir"], a["checkpoint_dir"], a["display_dir"]):
        os.makedirs(d, exist_ok=True)
    return argparse.Namespace(**a)
