"""GaussianNoiseLayer and the random-draw plumbing of the step (reference src/models/core/misc.py:18-26,
networks.py:130-135, adain_model.py:83-85).

The reference draws every random tensor on the CPU generator and copies it to the device.  Here the
large draw (the content-code noise, N x 256 x H/4 x W/4) is generated on the device by a Philox
counter kernel; the tiny ones ([N, 8]) come from torch's device generator.  For parity tests a
``ReplaySource`` injects the tensors recorded from a reference run instead.
"""
import torch
import torch.nn as nn

from ... import hip_ops as ops


class DeviceRandom:
    """Default source: noise generated on the GPU."""

    def __init__(self):
        self.offset = 0

    def add_noise(self, x):
        # seed from torch's CPU generator so torch.manual_seed() still makes runs reproducible
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return ops.gaussian_noise_add(x, seed, 0)

    def eps(self, shape, device):
        return torch.randn(shape, device=device)

    def z(self, shape, device):
        return torch.randn(shape, device=device)

    def dropout_mask(self, shape, device, keep=0.5):
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return ops.bernoulli_mask(shape, keep, seed, 0, device)


class ReplaySource:
    """Replays recorded draws in the reference's order."""

    def __init__(self, tensors):
        self.t = [torch.as_tensor(t) for t in tensors]
        self.i = 0

    def _next(self, shape, device):
        t = self.t[self.i]
        self.i += 1
        if tuple(t.shape) != tuple(shape):
            raise RuntimeError(f"replayed draw {self.i - 1} has shape {tuple(t.shape)}, step expects {tuple(shape)}")
        return t.to(device)

    def add_noise(self, x):
        return ops.add(x, ops.canon(self._next(x.shape, x.device)))

    eps = z = _next

    def dropout_mask(self, shape, device, keep=0.5):
        return ops.canon(self._next(shape, device))


_SOURCE = [DeviceRandom()]


def set_random_source(src):
    _SOURCE[0] = src if src is not None else DeviceRandom()


def random_source():
    return _SOURCE[0]


class GaussianNoiseLayer(nn.Module):
    def forward(self, x):
        if not self.training:
            return x
        return random_source().add_noise(x)



class Dropout(nn.Module):
    """nn.Dropout(0.5) of the decoders' residual blocks (--use_dropout): identity in eval mode."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p

    def forward(self, x):
        if not self.training:
            return x
        return ops.dropout(x, random_source().dropout_mask(x.shape, x.device, 1.0 - self.p), self.p)
