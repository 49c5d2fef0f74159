"""Composite blocks -- same class names, constructor signatures and state_dict keys as the reference's
``src/models/core/blocks.py`` (ConvBlock 10-46, UpsampleBlock 48-91, DownResnetBlock 93-119, ResnetBlock
121-138, AdaINResnetBlock 140-167, DecResnetBlock 169-208), but each block executes as a short chain of
fused HIP kernels: reflection padding lives inside the conv's gather, bias + activation in its epilogue,
normalisation + activation (+ the residual add) in one elementwise pass.

``nn.Conv2d`` / ``nn.ConvTranspose2d`` / ``nn.Linear`` objects are used purely as parameter holders (so
initialisation and checkpoint keys are the reference's); their own forward is never called.
"""
import torch
import torch.nn as nn

from ... import hip_ops as ops
from .functions import conv_weight, get_activation_layer, get_norm_layer, get_padding_layer, spectral_norm
from .misc import Dropout
from .norm import AdaptiveInstanceNorm, LayerNorm


class _Marker(nn.Module):
    """Parameter-free placeholder keeping the reference's nn.Sequential indices (pad / norm / act / pool)."""

    def __init__(self, kind, arg=None):
        super().__init__()
        self.kind, self.arg = kind, arg

    def extra_repr(self):
        return f"{self.kind}({self.arg})" if self.arg is not None else self.kind


def _make_norm(name, dim):
    if name == "layer":
        return LayerNorm(dim)
    if name == "instance":
        return _Marker("instance_norm", dim)
    if name == "batch":
        return nn.BatchNorm2d(dim)      # parameter / running-statistics holder (affine, momentum 0.1, eps 1e-5)
    raise NotImplementedError(f"norm '{name}' cannot be used inside a block")


def _batch_norm(bn, y, sums, act, training):
    """nn.BatchNorm2d semantics on the holder ``bn`` through the HIP kernels (running buffers updated in place)."""
    if training:
        bn.num_batches_tracked += 1
    return ops.batch_norm_act(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, training=training,
                              momentum=bn.momentum, act=act, eps=bn.eps, sums=sums)


class ConvBlock(nn.Module):
    """pad -> conv -> norm -> activation (reference blocks.py:10-46)."""

    def __init__(self, input_dim, output_dim, kernel_size, stride=1, padding=0, bias=False, norm_layer=None,
                 activation=None, padding_type=None, sn=False):
        super().__init__()
        self.pad_mode = "reflect" if get_padding_layer(padding_type) else "zero"
        self.act = get_activation_layer(activation)
        self.norm = get_norm_layer(norm_layer)
        self.stride, self.padding = stride, padding
        mods = []
        if self.pad_mode == "reflect":
            mods.append(_Marker("reflection_pad", padding))
        self._ci = len(mods)
        # padding of the holder is 0 when a pad layer precedes it, exactly like the reference
        conv = nn.Conv2d(input_dim, output_dim, kernel_size, stride, 0 if self.pad_mode == "reflect" else padding,
                         bias=bias)
        mods.append(spectral_norm(conv) if sn else conv)
        self._ni = None
        if self.norm is not None:
            self._ni = len(mods)
            mods.append(_make_norm(self.norm, output_dim))
        if self.act is not None:
            mods.append(_Marker(self.act))
        self.block = nn.Sequential(*mods)

    def forward(self, x, res=None, want_stats=False, grad_link=None, res_link=None, bwd_stats=None):
        """want_stats: return (conv output, its normalisation statistics) for an external norm layer (AdaIN).
        grad_link / res_link: the two ends of a residual block's ``ops.GradLink`` -- this block is the first convolution of
        the residual branch (its data gradient takes the skip gradient along), resp. the block whose norm adds ``res``."""
        conv = self.block[self._ci]
        weight = conv_weight(conv, self.training)
        if want_stats:
            assert self.norm is None and self.act is None and res is None
            return ops.conv2d(x, weight, conv.bias, stride=self.stride, pad=self.padding,
                              pad_mode=self.pad_mode, stats=True, grad_link=grad_link, bwd_stats=bwd_stats)
        if self.norm == "instance":
            # statistics come out of the GEMM epilogue; the bias in front of an affine-free InstanceNorm has an
            # identically zero gradient, so none is computed
            y, sums = ops.conv2d(x, weight, conv.bias, stride=self.stride, pad=self.padding,
                                 pad_mode=self.pad_mode, stats=True, bias_grad=False, grad_link=grad_link, bwd_stats=bwd_stats)
            return ops.instance_norm_act(y, act=self.act, res=res, sums=sums, res_link=res_link)
        assert grad_link is None and res_link is None and bwd_stats is None, \
            "residual gradient / statistics links are wired for the instance-norm blocks only"
        if self.norm == "batch":
            if self.training:
                y, sums = ops.conv2d(x, weight, conv.bias, stride=self.stride, pad=self.padding, pad_mode=self.pad_mode,
                                     stats=True)
            else:
                y, sums = ops.conv2d(x, weight, conv.bias, stride=self.stride, pad=self.padding,
                                     pad_mode=self.pad_mode), None
            y = _batch_norm(self.block[self._ni], y, sums, self.act, self.training)
            return y if res is None else ops.add(y, res)
        fused_act = self.act if self.norm is None else None
        y = ops.conv2d(x, weight, conv.bias, stride=self.stride, pad=self.padding, pad_mode=self.pad_mode,
                       act=fused_act)
        if self.norm == "layer":
            y = self.block[self._ni](y, act=self.act)
        return y if res is None else ops.add(y, res)


class UpsampleBlock(nn.Module):
    """up_type 'transpose': ConvTranspose2d -> norm -> activation; 'nearest': nn.Upsample(x2) -> ConvBlock (stride 1,
    zero padding -- the padding factory receives the already-resolved ``None``) -> norm -> activation
    (reference blocks.py:48-91; 'pixelshuffle' is not implemented)."""

    def __init__(self, input_dim, output_dim, kernel_size, stride=1, padding=0, output_padding=0, bias=False,
                 norm_layer=None, activation=None, padding_type=None, sn=False, up_type="transpose"):
        super().__init__()
        if sn:
            raise NotImplementedError("spectral norm in an UpsampleBlock: no reference model passes sn to a decoder "
                                      "(only --dis_sn exists, arguments.py:110)")
        self.act = get_activation_layer(activation)
        self.norm = get_norm_layer(norm_layer)
        self.stride, self.padding, self.output_padding = stride, padding, output_padding
        self.nearest = "transpose" not in up_type
        if "transpose" in up_type:
            mods = [nn.ConvTranspose2d(input_dim, output_dim, kernel_size, stride, padding, output_padding, bias=bias)]
        elif "nearest" in up_type:
            get_padding_layer(padding_type)       # (raises for unsupported types like the reference)
            mods = [_Marker("upsample_nearest", 2),
                    ConvBlock(input_dim, output_dim, kernel_size, 1, padding, padding_type=None, bias=bias, sn=sn)]
        else:
            raise NotImplementedError(f"Mode {up_type} is not supported at the moment")
        self._ni = None
        if self.norm is not None:
            self._ni = len(mods)
            mods.append(_make_norm(self.norm, output_dim))
        if self.act is not None:
            mods.append(_Marker(self.act))
        self.block = nn.Sequential(*mods)

    def forward(self, x):
        fused_act = self.act if self.norm is None else None
        if self.nearest:
            inner = self.block[1]
            conv = inner.block[inner._ci]
            y = ops.conv2d(ops.upsample2_nearest(x), conv.weight, conv.bias, stride=1, pad=inner.padding,
                           pad_mode="zero", act=fused_act)
        else:
            conv = self.block[0]
            y = ops.conv_transpose2d(x, conv.weight, conv.bias, stride=self.stride, pad=self.padding,
                                     out_pad=self.output_padding, act=fused_act)
        if self.norm == "instance":
            return ops.instance_norm_act(y, act=self.act)
        if self.norm == "layer":
            return self.block[self._ni](y, act=self.act)
        if self.norm == "batch":
            return _batch_norm(self.block[self._ni], y, None, self.act, self.training)
        return y


class DownResnetBlock(nn.Module):
    """reference blocks.py:93-119.  The first layer of ``conv`` is an in-place activation of the block
    input, so the shortcut branch consumes the ACTIVATED tensor too."""

    def __init__(self, input_dim, output_dim, norm_layer="instance", activation="lrelu", padding_type="reflect",
                 bias=True):
        super().__init__()
        if norm_layer is not None:
            raise NotImplementedError("DownResnetBlock with a norm layer is not on the hot path (AdaINModel passes None)")
        self.act = get_activation_layer(activation)
        self.conv = nn.Sequential(
            _Marker(self.act),
            ConvBlock(input_dim, input_dim, 3, 1, padding=1, padding_type=padding_type, norm_layer=None,
                      activation=activation, bias=bias),
            ConvBlock(input_dim, output_dim, 3, 1, padding=1, padding_type=padding_type, bias=bias),
            _Marker("avgpool2"))
        self.shortcut = nn.Sequential(_Marker("avgpool2"), nn.Conv2d(input_dim, output_dim, 1, 1, 0, bias=bias))

    def forward(self, x):
        a = ops.activation(x, self.act)
        h = ops.avg_pool2(self.conv[2](self.conv[1](a)))
        sc = self.shortcut[1]
        s = ops.conv2d(ops.avg_pool2(a), sc.weight, sc.bias)
        return ops.add(h, s)


class ResnetBlock(nn.Module):
    """x + [conv-IN-act, conv-IN](x) (reference blocks.py:121-138); the add rides on the second norm pass."""

    def __init__(self, input_dim, output_dim, dropout=False, norm_layer="instance", padding_type="reflect",
                 activation="relu"):
        super().__init__()
        mods = [ConvBlock(input_dim, output_dim, 3, 1, 1, padding_type=padding_type, norm_layer=norm_layer,
                          activation=activation),
                ConvBlock(output_dim, output_dim, 3, 1, 1, padding_type=padding_type, norm_layer=norm_layer)]
        if dropout:
            mods.append(Dropout(0.5))           # (index 2 of the Sequential, like the reference's nn.Dropout)
        self.model = nn.Sequential(*mods)

    def forward(self, x):
        if len(self.model) > 2 and self.training:
            return ops.add(self.model[2](self.model[1](self.model[0](x))), x)
        if self.model[0].norm == "instance" and self.model[1].norm == "instance":
            # the skip gradient rides on the first convolution's data gradient instead of autograd's accumulation pass
            link = ops.GradLink()
            # x's gradient is formed entirely in conv1's data-gradient epilogue (dgrad + skip): if a normalisation produced
            # x, its backward statistics come out of that epilogue too; likewise conv2 for the norm in between
            h = self.model[0](x, grad_link=link, bwd_stats=ops.stats_link_of(x))
            return self.model[1](h, res=x, res_link=link, bwd_stats=ops.stats_link_of(h))
        return self.model[1](self.model[0](x), res=x)


class AdaINResnetBlock(nn.Module):
    """conv-AdaIN-act-conv-AdaIN + residual with ONE shared AdaIN projection (reference blocks.py:140-167)."""

    def __init__(self, input_dim, output_dim, dropout=False, style_dim=256, padding_type="reflect", activation="relu"):
        super().__init__()
        self.act = get_activation_layer(activation)
        self.activation = _Marker(self.act)
        self.conv1 = ConvBlock(input_dim, output_dim, 3, 1, 1, padding_type=padding_type)
        self.conv2 = ConvBlock(output_dim, output_dim, 3, 1, 1, padding_type=padding_type)
        self.norm = AdaptiveInstanceNorm(output_dim, style_dim)
        self.dropout = Dropout(0.5) if dropout else nn.Identity()

    def forward(self, x, z, gb=None):
        # the reference evaluates norm.fc(z) at both norm sites (blocks.py:158-164): same weights, same input, so
        # one projection serves both (its gradient is the sum of the two sites'); the decoder may hand in the
        # projection it computed for all of its blocks in one launch
        if gb is None:
            gb = self.norm.project(z)
        plain = not (isinstance(self.dropout, Dropout) and self.training)
        link = ops.GradLink() if plain else None         # (the skip gradient rides on conv1's data gradient)
        y, sums = self.conv1(x, want_stats=True, grad_link=link, bwd_stats=ops.stats_link_of(x) if plain else None)
        h = self.norm(y, z, act=self.act, sums=sums, gb=gb)
        y, sums = self.conv2(h, want_stats=True, bwd_stats=ops.stats_link_of(h))
        if not plain:
            return ops.add(self.dropout(self.norm(y, z, sums=sums, gb=gb)), x)     # (the add cannot ride on the norm pass)
        return self.norm(y, z, res=x, sums=sums, gb=gb, res_link=link)


def _expand_planes(v, ref):
    return v.view(v.size(0), v.size(1), 1, 1).expand(v.size(0), v.size(1), ref.size(2), ref.size(3))


class DecResnetBlock(nn.Module):
    """BaseModel decoder block (reference blocks.py:169-208); the final add is out of place (the reference's
    in-place ``out += residual`` on a ReLU output raises under current autograd)."""

    def __init__(self, n_channel, add_channel, norm_layer="instance", padding_type="reflect", stride=1, dropout=False):
        super().__init__()
        self.conv1 = ConvBlock(n_channel, n_channel, 3, stride=stride, padding=1, padding_type=padding_type)
        self.conv2 = ConvBlock(n_channel, n_channel, 3, stride=stride, padding=1, padding_type=padding_type)
        self.norm = _Marker("instance_norm", n_channel)
        nca = n_channel + add_channel
        self.block1 = nn.Sequential(nn.Conv2d(nca, nca, 1), _Marker("relu"), nn.Conv2d(nca, n_channel, 1), _Marker("relu"))
        self.block2 = nn.Sequential(nn.Conv2d(nca, nca, 1), _Marker("relu"), nn.Conv2d(nca, n_channel, 1), _Marker("relu"))
        self.dropout = Dropout(0.5) if dropout else nn.Identity()

    @staticmethod
    def _mix(block, t):
        t = ops.conv2d(t, block[0].weight, block[0].bias, act="relu")
        return ops.conv2d(t, block[2].weight, block[2].bias, act="relu")

    def forward(self, x, z):
        ze = _expand_planes(z, x).to(ops.compute_dtype())
        out = ops.instance_norm_act(self.conv1(x))
        out = self._mix(self.block1, torch.cat([out, ze], dim=1))
        out = ops.instance_norm_act(self.conv2(out))
        out = self._mix(self.block2, torch.cat([out, ze], dim=1))
        return ops.add(self.dropout(out), x)
