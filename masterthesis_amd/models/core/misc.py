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
    """Default source: noise generated on the GPU.  The generator state {seed, draw counter} of the Philox kernels
    lives in device memory: no random value is passed from the host per launch, so a whole training step can be
    captured into a hipGraph and still draw fresh noise on every replay.  The seed is taken ONCE from torch's CPU
    generator, so ``torch.manual_seed`` (and the per-rank reseeding of a multi-GPU run) still decide the stream."""

    def __init__(self):
        self._state = {}

    def state(self, device):
        dev = torch.device(device)
        if dev.index is None and dev.type == "cuda":
            dev = torch.device("cuda", torch.cuda.current_device())
        st = self._state.get(dev)
        if st is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            st = torch.tensor([seed, 0], dtype=torch.int64).to(dev)
            self._state[dev] = st
        return st

    def add_noise(self, x):
        return ops.gaussian_noise_add_dev(x, self.state(x.device))

    def eps(self, shape, device):
        return torch.randn(shape, device=device)        # (torch's device generator is hipGraph-safe as well)

    def z(self, shape, device):
        return torch.randn(shape, device=device)

    def dropout_mask(self, shape, device, keep=0.5):
        return ops.bernoulli_mask_dev(shape, keep, self.state(device), device)


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
