"""Training-harness base class -- mirror of the reference's ``src/models/model.py:15-122``: networks,
optimizers, schedulers and losses live in AttributeDicts (NOT registered sub-modules, exactly like the
reference), ``initialize / update_lr / save / load / save_images / write_loss / print_losses``.

Checkpoints keep the reference format: ``model_{it}.ckpt`` = {net: state_dict}, ``opt_{it}.ckpt`` =
{net: Adam state_dict}; ``module.``-prefixed keys written by the reference's DataParallel runs load too.
"""
import json
import os
from abc import ABC, abstractmethod

import torch
import torch.nn as nn

from .. import hip_ops as ops
from ..distributed import GradReducer, broadcast_parameters, seed_rank_streams
from ..utils import AttributeDict, save_image_grid
from .core.functions import get_scheduler, init_net


class _JsonlWriter:
    """Scalar logger used when tensorboardX is not installed (it is not in this image)."""

    def __init__(self, log_dir):
        os.makedirs(log_dir, exist_ok=True)
        self.path = os.path.join(log_dir, "scalars.jsonl")

    def add_scalar(self, tag, value, step):
        with open(self.path, "a") as f:
            f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")


def _make_writer(log_dir):
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter(log_dir=log_dir)
    except Exception:
        return _JsonlWriter(log_dir)


class Model(ABC, nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.device = "cuda" if torch.cuda.is_available() else "cpu"
        object.__setattr__(self, "model", AttributeDict())
        object.__setattr__(self, "optimizer", AttributeDict())
        object.__setattr__(self, "scheduler", AttributeDict())
        object.__setattr__(self, "loss", AttributeDict())
        self._loss_t = {}
        precision = getattr(args, "precision", "fp32")
        if precision not in ("fp32", "bf16"):
            raise ValueError(f"--precision must be fp32 or bf16, got {precision}")
        ops.set_compute_dtype(torch.bfloat16 if precision == "bf16" else torch.float32)
        if "train" in args.mode:
            self.writer = _make_writer(args.logdir)
        self.print_loss = []
        self.reducer = GradReducer()

    @abstractmethod
    def set_inputs(self, inputs):
        """set batch inputs"""

    @abstractmethod
    def optimize_parameters(self, global_iter):
        """one training iteration"""

    def initialize(self):
        init_type = None if self.args.resume else self.args.init_type
        for net in self.model:
            self.model[net] = init_net(self.model[net], init_type=init_type, gpu_ids=self.args.gpu_ids,
                                       device=self.device)
        if "train" in self.args.mode:
            self.args.last_iter = -1 if self.args.resume_opt is None else self.args.last_iter
            self.init_scheduler()
            self.load(self.args.resume, self.args.resume_opt)
        else:
            self.load(self.args.resume)
        # replicas start from rank 0's weights AND buffers (spectral-norm u / v, BatchNorm running statistics) ...
        broadcast_parameters([p.data for net in self.model for p in self.model[net].parameters()] +
                             [b.data for net in self.model for b in self.model[net].buffers()])
        # ... but draw their own noise / eps / z_random / dropout masks
        self.rng_seed = seed_rank_streams()

    def init_scheduler(self):
        for opt in self.optimizer:
            # resuming (last_iter >= 0): torch's schedulers require 'initial_lr'; the reference creates the
            # scheduler BEFORE loading the optimizer checkpoint (model.py:47-52) and trips over exactly that
            for group in self.optimizer[opt].param_groups:
                group.setdefault("initial_lr", group["lr"])
            self.scheduler[opt] = get_scheduler(self.optimizer[opt], self.args, self.args.last_iter)

    def get_current_lr(self):
        return {opt: self.optimizer[opt].param_groups[0]["lr"] for opt in self.optimizer}

    def update_lr(self):
        for net in self.model:
            if net in self.scheduler:
                self.scheduler[net].step()

    def save(self, it):
        # --ckpt_module_prefix: write the keys the reference's default GPU run expects -- its init_net wraps every
        # network in nn.DataParallel (functions.py:98-101), so its strict load_state_dict wants 'module.'-prefixed keys
        pre = "module." if getattr(self.args, "ckpt_module_prefix", False) else ""
        model_state = {net: {pre + k: v for k, v in self.model[net].state_dict().items()} for net in self.model}
        torch.save(model_state, os.path.join(self.args.checkpoint_dir, f"model_{it}.ckpt"))
        opt_state = {opt: self.optimizer[opt].state_dict() for opt in self.optimizer}
        torch.save(opt_state, os.path.join(self.args.checkpoint_dir, f"opt_{it}.ckpt"))

    @staticmethod
    def _strip_module_prefix(sd):
        if sd and all(k.startswith("module.") for k in sd):
            return {k[len("module."):]: v for k, v in sd.items()}
        return sd

    def load(self, checkpoint, opt_ckpt=None):
        if checkpoint is not None:
            ckpt = torch.load(checkpoint, map_location="cpu")
            for net in ckpt:
                if net in self.model.keys():
                    print(f"Loading checkpoint for : {net}")
                    self.model[net].load_state_dict(self._strip_module_prefix(ckpt[net]))
                else:
                    print(f"Checkpoint for {net} network is not found.")
        if opt_ckpt is not None:
            ckpt = torch.load(opt_ckpt, map_location="cpu")
            for opt in ckpt:
                if opt in self.optimizer.keys():
                    print(f"Loading checkpoint for {opt} optimizer.")
                    self.optimizer[opt].load_state_dict(ckpt[opt])
                else:
                    print(f"Checkpoint for {opt} optimizer is not found.")

    def save_images(self, it):
        visuals = ops.to_nchw_f32(self.compute_visuals()).cpu()
        save_image_grid(visuals / 2 + 0.5, os.path.join(self.args.display_dir, f"gen_{it}.jpg"))

    # losses are kept as device scalars during the step (no host sync); floats are produced on demand
    def _set_loss(self, **kw):
        for k, v in kw.items():
            self._loss_t[k] = v.detach() if torch.is_tensor(v) else v

    def sync_losses(self):
        if self._loss_t:
            keys = list(self._loss_t)
            vals = [torch.as_tensor(self._loss_t[k], dtype=torch.float32, device=self.device).reshape(()) for k in keys]
            # the device status words (kernels that can fail on the device set a bit instead of failing silently) ride on the
            # same device->host copy: a failed step raises here, at the first place the host looks at the device's scalars
            dev = torch.device(self.device)
            if dev.type == "cuda" and dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            st = ops.device_status(dev) if dev.type == "cuda" else None
            flat = torch.stack(vals)
            if st is not None:
                flat = torch.cat([flat, st[:2].to(torch.float32)])
            host = flat.cpu().tolist()                              # ONE device->host copy
            if st is not None:
                ops.raise_on_device_status([int(v) for v in host[len(keys):]], dev)
            for k, v in zip(keys, host):
                self.loss[k] = v
        return self.loss

    def write_loss(self, global_iter):
        self.sync_losses()
        for name in self.loss:
            self.writer.add_scalar(name, self.loss[name], global_iter)

    def print_losses(self):
        self.sync_losses()
        return {k: self.loss[k] for k in self.loss if k in self.print_loss}

    def compute_metrics(self):
        pass
