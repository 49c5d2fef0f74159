"""Network definitions of the hot path -- class names, constructor signatures and state_dict keys of
the reference's ``src/models/core/networks.py`` (ContentEncoder 8-43, StyleEncoder 45-85,
ReparameterizedStyleEncoder 87-146, Decoder 148-205, AdaINDecoder 207-270, DecoderConcat 272-333,
Discriminator 335-384, ContentDiscriminator 386-419, MultiScaleDiscriminator 421-466).

Inputs may be ordinary NCHW fp32 tensors; everything downstream of the first op is a canonical
padded-NHWC activation (see ``hip_ops``).  ``ResnetGenerator`` (468-512) is unused by the reference's
models and not provided.
"""
import numpy as np
import torch
import torch.nn as nn

from ... import hip_ops as ops
from .blocks import (AdaINResnetBlock, ConvBlock, DecResnetBlock, DownResnetBlock, ResnetBlock, UpsampleBlock,
                     _Marker, _expand_planes)
from .functions import conv_weight, get_activation_layer
from .misc import GaussianNoiseLayer, random_source


class ContentEncoder(nn.Module):
    def __init__(self, input_dim, dim=64, num_downs=2, n_blocks=4, norm_layer="instance", padding_type="reflect",
                 bias=True):
        super().__init__()
        layers = [ConvBlock(input_dim, dim, 7, 1, 3, padding_type=padding_type, norm_layer=norm_layer,
                            activation="lrelu", bias=bias)]
        for _ in range(num_downs):
            layers.append(ConvBlock(dim, dim * 2, 3, 2, 1, padding_type=padding_type, norm_layer=norm_layer,
                                    activation="relu", bias=bias))
            dim *= 2
        for _ in range(n_blocks):
            layers.append(ResnetBlock(dim, dim, norm_layer=norm_layer, activation="relu"))
        layers.append(GaussianNoiseLayer())
        self.model = nn.Sequential(*layers)
        self.output_dim = dim

    def features(self, x):
        """everything in front of the GaussianNoiseLayer: a deterministic function of (x, weights)"""
        for layer in self.model[:-1]:
            x = layer(x)
        return x

    def add_noise(self, h):
        return self.model[-1](h)

    def forward(self, x):
        return self.add_noise(self.features(x))


class StyleEncoder(nn.Module):
    """BaseModel without --reparam (reference networks.py:45-85)."""

    def __init__(self, input_dim, output_dim=8, dim=64, num_downs=4, num_domains=2, padding_type="reflect",
                 activation="relu"):
        super().__init__()
        layers = [ConvBlock(input_dim + num_domains, dim, 7, 1, padding=3, padding_type=padding_type,
                            activation=activation)]
        max_filter_size = 256
        for _ in range(num_downs):
            in_dim, out_dim = min(max_filter_size, dim), min(max_filter_size, dim * 2)
            layers.append(ConvBlock(in_dim, out_dim, 4, 2, padding=1, padding_type=padding_type, activation=activation))
            dim *= 2
        layers.append(_Marker("adaptive_avgpool"))
        layers.append(nn.Conv2d(out_dim, output_dim, 1, 1, 0))
        self.model = nn.Sequential(*layers)

    def forward(self, x, c):
        h = ops.cat_class_planes(x, c)
        for layer in self.model[:-2]:
            h = layer(h)
        head = self.model[-1]
        return ops.linear(ops.global_avg_pool(h), head.weight.flatten(1), head.bias)


class ReparameterizedStyleEncoder(nn.Module):
    def __init__(self, input_dim, output_dim=8, dim=64, n_blocks=4, num_domains=2, norm_layer=None, activation=None,
                 bias=True):
        super().__init__()
        self.act = get_activation_layer(activation)
        if self.act is None:
            raise ValueError("ReparameterizedStyleEncoder needs an activation (the reference fails on None as well)")
        max_filter_size = 256
        layers = [ConvBlock(input_dim + num_domains, dim, 4, 2, 1, padding_type="reflect", bias=bias)]
        for _ in range(1, n_blocks):
            in_dim, out_dim = min(max_filter_size, dim), min(max_filter_size, dim * 2)
            layers.append(DownResnetBlock(in_dim, out_dim, norm_layer, activation, bias=bias))
            dim *= 2
        layers.append(_Marker(self.act))
        layers.append(_Marker("adaptive_avgpool"))
        self.model = nn.Sequential(*layers)
        self.out_nch = out_dim
        self.fc = nn.Linear(out_dim, output_dim)
        self.fcVar = nn.Linear(out_dim, output_dim)

    def reparameterize(self, mu, logvar):
        eps = random_source().eps(tuple(mu.shape), mu.device)
        return ops.reparameterize(mu, logvar, eps)

    def moments(self, x, c):
        """(mu, logvar): the deterministic part of the encoder (everything in front of the eps draw)"""
        h = ops.cat_class_planes(x, c)
        for layer in self.model[:-2]:
            h = layer(h)
        flat = ops.global_avg_pool(ops.activation(h, self.act))
        mu = ops.linear(flat, self.fc.weight, self.fc.bias)
        logvar = ops.linear(flat, self.fcVar.weight, self.fcVar.bias)
        return mu, logvar

    def forward(self, x, c):
        mu, logvar = self.moments(x, c)
        return self.reparameterize(mu, logvar), mu, logvar


def _style_mlp(latent_dim, num_domains, out_dim):
    return nn.Sequential(nn.Linear(latent_dim + num_domains, 256), _Marker("relu"), nn.Linear(256, 256),
                         _Marker("relu"), nn.Linear(256, out_dim))


def _run_mlp(mlp, x):
    h = ops.activation(ops.linear(x, mlp[0].weight, mlp[0].bias), "relu")
    h = ops.activation(ops.linear(h, mlp[2].weight, mlp[2].bias), "relu")
    return ops.linear(h, mlp[4].weight, mlp[4].bias)


def _upsampling(dim, output_dim, num_ups, up_type, norm_layer, activation, bias):
    ups = []
    for _ in range(num_ups):
        ups.append(UpsampleBlock(dim, dim // 2, 3, 2, 1, 1, norm_layer=norm_layer, activation=activation,
                                 up_type=up_type, bias=bias))
        dim = dim // 2
    ups.append(_to_rgb(dim, output_dim, up_type))
    return nn.Sequential(*ups)


def _to_rgb(dim, output_dim, up_type):
    """last decoder layer (networks.py:185-188): 1x1 ConvTranspose + tanh, or 7x7 zero-padded conv + tanh for the
    non-transpose up-sampling types"""
    if "transpose" in up_type:
        return UpsampleBlock(dim, output_dim, 1, 1, 0, activation="tanh", up_type="transpose")
    return ConvBlock(dim, output_dim, 7, 1, 3, activation="tanh")


class Decoder(nn.Module):
    """BaseModel without --concat (reference networks.py:148-205)."""

    def __init__(self, output_dim, dim=256, n_blocks=4, num_domains=2, num_ups=2, latent_dim=8, up_type="transpose",
                 dropout=False, norm_layer="layer", activation="relu", bias=True):
        super().__init__()
        self.dim_add = dim
        self.dec1 = nn.ModuleList([DecResnetBlock(dim, self.dim_add, dropout=dropout) for _ in range(n_blocks)])
        self.dec2 = _upsampling(dim, output_dim, num_ups, up_type, norm_layer, activation, bias)
        self.linear = _style_mlp(latent_dim, num_domains, self.dim_add * n_blocks)

    def forward(self, x, z, c):
        z_c = _run_mlp(self.linear, torch.cat([c, z], 1))
        out = x
        for dec, zi in zip(self.dec1, torch.split(z_c, self.dim_add, dim=1)):
            out = dec(out, zi.contiguous())
        for up in self.dec2:
            out = up(out)
        return out


class AdaINDecoder(nn.Module):
    def __init__(self, output_dim, dim=256, n_blocks=4, num_domains=2, num_ups=2, latent_dim=8, up_type="transpose",
                 res_norm="adain", dropout=False, norm_layer="layer", activation="relu", bias=True):
        super().__init__()
        if "adain" not in res_norm:
            raise NotImplementedError("AdaINDecoder without adain residual blocks is unused by the reference models")
        self.dim_add = dim
        self.dec1 = nn.ModuleList([AdaINResnetBlock(dim, self.dim_add, style_dim=self.dim_add, dropout=dropout)
                                   for _ in range(n_blocks)])
        self.dec2 = _upsampling(dim, output_dim, num_ups, up_type, norm_layer, activation, bias)
        self.linear = _style_mlp(latent_dim, num_domains, self.dim_add)

    def forward(self, x, z, c):
        z_c = _run_mlp(self.linear, torch.cat([c, z], 1))      # class first (networks.py:264)
        # every block projects the SAME style code with its own norm.fc (blocks.py:152): all projections in one launch
        gbs = ops.linear_grouped(z_c, [(dec.norm.fc.weight, dec.norm.fc.bias) for dec in self.dec1])
        out = x
        for dec, gb in zip(self.dec1, gbs):
            out = dec(out, z_c, gb=gb)
        for up in self.dec2:
            out = up(out)
        return out


class DecoderConcat(nn.Module):
    """BaseModel --concat (reference networks.py:272-333): class / style planes concatenated on channels."""

    def __init__(self, output_dim, dim=256, n_blocks=3, num_domains=2, latent_dim=8, up_type="transpose", dropout=False,
                 norm_layer="layer", activation="relu", bias=True):
        super().__init__()
        self.dec_share = ResnetBlock(dim, dim)
        nch = dim + latent_dim + num_domains
        self.dec1 = nn.Sequential(*[ResnetBlock(nch, nch, dropout=dropout) for _ in range(n_blocks)])
        nch = nch + latent_dim
        self.dec2 = UpsampleBlock(nch, nch // 2, 3, 2, 1, 1, norm_layer=norm_layer, activation=activation,
                                  up_type=up_type, bias=bias)
        nch = nch // 2 + latent_dim
        self.dec3 = UpsampleBlock(nch, nch // 2, 3, 2, 1, 1, norm_layer=norm_layer, activation=activation,
                                  up_type=up_type, bias=bias)
        nch = nch // 2 + latent_dim
        self.dec4 = _to_rgb(nch, output_dim, up_type)

    def forward(self, x, z, c):
        dt = ops.compute_dtype()

        def planes(v, ref):
            return _expand_planes(v, ref).to(dt)
        out0 = self.dec_share(x)
        h = torch.cat([out0, planes(c, out0), planes(z, out0)], 1)
        for blk in self.dec1:
            h = blk(h)
        h = self.dec2(torch.cat([h, planes(z, h)], 1))
        h = self.dec3(torch.cat([h, planes(z, h)], 1))
        return self.dec4(torch.cat([h, planes(z, h)], 1))


class Discriminator(nn.Module):
    def __init__(self, input_dim, dim=64, n_layers=6, num_domains=2, norm_layer=None, activation="lrelu",
                 padding_type="reflect", bias=True, sn=False, image_size=256):
        super().__init__()
        layers = [ConvBlock(input_dim, dim, kernel_size=3, stride=2, padding=1, padding_type=padding_type,
                            norm_layer=norm_layer, sn=sn, activation=activation, bias=bias)]
        nch = dim
        for _ in range(n_layers - 2):
            layers.append(ConvBlock(nch, nch * 2, kernel_size=3, stride=2, padding=1, padding_type=padding_type,
                                    norm_layer=norm_layer, sn=sn, activation=activation, bias=bias))
            nch *= 2
        layers.append(ConvBlock(nch, nch, kernel_size=3, stride=2, padding=1, padding_type=padding_type, sn=sn,
                                activation=activation, bias=bias))
        self.model = nn.Sequential(*layers)
        self.conv1 = nn.Conv2d(nch, 1, kernel_size=1, stride=1, padding=1, bias=False)
        kernel = int(image_size / np.power(2, n_layers))
        self.conv2 = nn.Conv2d(nch, num_domains, kernel_size=kernel, bias=False)
        self.pool = _Marker("adaptive_avgpool")
        self.output_dim = nch

    def forward(self, x):
        h = x
        for layer in self.model:
            h = layer(h)
        out = ops.conv2d(h, self.conv1.weight, None, stride=1, pad=1)        # 1x1 with padding=1 (networks.py:373)
        out_cls = ops.global_avg_pool(ops.conv2d(h, self.conv2.weight, None))
        return out, out_cls


class ContentDiscriminator(nn.Module):
    def __init__(self, dim=256, num_domains=3, norm_layer="instance", activation="lrelu", padding_type="reflect",
                 bias=True):
        super().__init__()
        layers = [ConvBlock(dim, dim, kernel_size=7, stride=2, padding=1, padding_type=padding_type,
                            norm_layer=norm_layer, activation=activation, bias=bias) for _ in range(3)]
        layers.append(ConvBlock(dim, dim, kernel_size=4, stride=1, padding=0, padding_type=padding_type,
                                activation=activation, bias=bias))
        layers.append(nn.Conv2d(dim, num_domains, kernel_size=1, stride=1, padding=0))
        self.pool = _Marker("adaptive_avgpool")
        self.model = nn.Sequential(*layers)

    def forward(self, x):
        h = x
        for layer in self.model[:-1]:
            h = layer(h)
        head = self.model[-1]
        return ops.global_avg_pool(ops.conv2d(h, head.weight, head.bias))


_MSD_MERGE_FACTOR = float(__import__("os").environ.get("MT_MSD_MERGE_FACTOR", "25"))   # patch elements <= factor x weight elements
_MSD_MERGE = [__import__("os").environ.get("MT_MSD_MERGE", "1") != "0"]
_MSD_LAYER_MAJOR = [__import__("os").environ.get("MT_MSD_LAYER_MAJOR", "1") != "0"]


class MultiScaleDiscriminator(nn.Module):
    def __init__(self, input_dim, dim=64, n_layers=6, num_domains=2, norm_layer=None, activation="lrelu",
                 padding_type=None, num_scales=3, sn=False):
        super().__init__()
        self.num_scales = num_scales
        self.downsample = _Marker("avgpool3s2")
        layers = [ConvBlock(input_dim, dim, 4, 2, 1, norm_layer=None, activation=activation, padding_type=padding_type,
                            sn=sn)]
        for _ in range(n_layers - 1):
            layers.append(ConvBlock(dim, dim * 2, 4, 2, 1, norm_layer=norm_layer, activation=activation,
                                    padding_type=padding_type, sn=sn))
            dim *= 2
        self.model = nn.Sequential(*layers)
        self.dis = nn.Conv2d(dim, 1, 1, 1, 0)
        self.cls = nn.Conv2d(dim, num_domains, 1, 1, 0)
        self.pool = _Marker("adaptive_avgpool")

    @staticmethod
    def _mergeable(layer, hs):
        """A deep layer whose time is streaming its weights (1024 -> 2048: 67 MB of bf16 per launch, three launches for three
        scales) and whose patches are small next to them: its scales run as ONE batch of 4x4 mini-images (ops.patch4s2_multi)
        through the same weights as a 4x4 / stride 4 convolution -- same products, one pass over the weights, one weight
        gradient.  Not with spectral norm (its power iteration runs once per CALL in the reference) or a norm layer."""
        if not (_MSD_MERGE[0] and len(hs) > 1 and isinstance(layer, ConvBlock) and hs[0].is_cuda):
            return False
        conv = layer.block[layer._ci]
        if (not isinstance(conv, nn.Conv2d) or getattr(conv, "_mt_sn", None) is not None or layer.norm is not None
                or layer.pad_mode != "zero"):
            return False
        if conv.kernel_size != (4, 4) or layer.stride != 2 or layer.padding != 1:
            return False
        if any(h.shape[2] % 2 or h.shape[3] % 2 for h in hs):
            return False
        ci, co = conv.in_channels, conv.out_channels
        patch_elems = sum(h.shape[0] * (h.shape[2] // 2) * (h.shape[3] // 2) for h in hs) * 16 * ci
        return patch_elems <= _MSD_MERGE_FACTOR * co * ci * 16 and ci % 8 == 0

    @staticmethod
    def _merged(layer, hs):
        conv = layer.block[layer._ci]
        col = ops.patch4s2_multi(hs)
        y = ops.conv2d(col, conv_weight(conv, layer.training), conv.bias, stride=4, pad=0, pad_mode="zero", act=layer.act)
        return ops.split_pixels(y, [(h.shape[0], h.shape[2] // 2, h.shape[3] // 2) for h in hs])

    def forward(self, x):
        if not _MSD_LAYER_MAJOR[0]:
            outputs = []
            for _ in range(self.num_scales):
                h = x
                for layer in self.model:
                    h = layer(h)
                dis = ops.conv2d(h, self.dis.weight, self.dis.bias)
                c = ops.global_avg_pool(ops.conv2d(h, self.cls.weight, self.cls.bias))
                outputs.append((dis, c))
                x = ops.avg_pool3s2(x)
            return outputs
        # layer-major order (same operations, same results): the scales share the weights, and the deep layers are bound by
        # streaming them (1024 -> 2048: 67 MB of bf16 per launch) -- the three uses of a layer's weights run back to back while
        # the Infinity Cache still holds them (autograd runs the backward in the mirrored order)
        hs = [x]
        for _ in range(self.num_scales - 1):
            hs.append(ops.avg_pool3s2(hs[-1]))
        for layer in self.model:
            hs = self._merged(layer, hs) if self._mergeable(layer, hs) else [layer(h) for h in hs]
        return [(ops.conv2d(h, self.dis.weight, self.dis.bias),
                 ops.global_avg_pool(ops.conv2d(h, self.cls.weight, self.cls.bias))) for h in hs]
