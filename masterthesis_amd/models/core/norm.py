"""LayerNorm / AdaptiveInstanceNorm parameter holders (reference src/models/core/norm.py:5-33).
The arithmetic (statistics + affine + activation [+ residual]) runs in the fused HIP kernels behind
``hip_ops.layer_norm_act`` / ``hip_ops.adain_act``."""
import torch
import torch.nn as nn

from ... import hip_ops as ops


class LayerNorm(nn.Module):
    """Per-sample normalisation over (C,H,W) with per-channel affine; eps is fixed at 1e-5 like the
    reference (its ``eps`` ctor argument is ignored there too, norm.py:6-21)."""

    def __init__(self, n_out, eps=1e-5, affine=True):
        super().__init__()
        self.n_out = n_out
        self.affine = affine
        if affine:
            self.weight = nn.Parameter(torch.ones(n_out, 1, 1))
            self.bias = nn.Parameter(torch.zeros(n_out, 1, 1))

    def forward(self, x, act=None):
        if self.affine:
            return ops.layer_norm_act(x, self.weight, self.bias, act=act)
        return ops.layer_norm_act(x, None, None, act=act)


class AdaptiveInstanceNorm(nn.Module):
    """(1 + w) * IN(x) + b with [w, b] = fc(s) (norm.py:23-33)."""

    def __init__(self, num_features, style_dim):
        super().__init__()
        self.num_features = num_features
        self.fc = nn.Linear(style_dim, num_features * 2)

    def project(self, s):
        """[w, b] = fc(s); a block that normalises twice with the same style (blocks.py:152-164) projects once"""
        return ops.linear(s, self.fc.weight, self.fc.bias)

    def forward(self, x, s, act=None, res=None, sums=None, gb=None, res_link=None):
        h = self.project(s) if gb is None else gb
        return ops.adain_act(x, h, act=act, res=res, sums=sums, res_link=res_link)
