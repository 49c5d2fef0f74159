"""Factories, weight init and LR schedulers -- host-side mirror of the reference's
``src/models/core/functions.py`` (names, argument meaning and error behaviour kept).

Differences forced by the MI355X design: norm / activation / padding "layers" are lightweight
spec objects (the arithmetic is fused into HIP kernels by ``blocks.ConvBlock``), and multi-GPU is one
process per GPU with RCCL all-reduce (``masterthesis_amd.distributed``) instead of the reference's
single-process ``nn.DataParallel`` (functions.py:96-106).
"""
import torch
import torch.nn as nn
from torch.nn import init
from torch.optim import lr_scheduler

NORMS = ("batch", "instance", "layer", "adain")
ACTIVATIONS = ("relu", "lrelu", "tanh", "sigmoid")
PADDINGS = ("reflect", "replicate")


def get_norm_layer(norm_layer="instance"):
    """Validate a norm name (reference functions.py:11-26).  Returns the name or None."""
    if norm_layer is None:
        return None
    if not isinstance(norm_layer, str):
        raise ValueError(f"parameter type of norm_layer should be 'str', but got {norm_layer}.")
    if norm_layer not in NORMS:
        raise NotImplementedError(f"norm type '{norm_layer}' is not supported at the moment")
    return norm_layer


def get_activation_layer(activation=None):
    """Validate an activation name (reference functions.py:28-43)."""
    if activation is None:
        return None
    if not isinstance(activation, str):
        raise ValueError(f"parameter type of activation should be 'str', but got {type(activation)}.")
    if activation not in ACTIVATIONS:
        raise NotImplementedError(f"activation type '{activation}' is not supported at the moment")
    if activation == "sigmoid":
        raise NotImplementedError("activation 'sigmoid' is unused on the hot path and has no HIP kernel")
    return activation


def get_padding_layer(padding_type=None):
    """Validate a padding name (reference functions.py:45-58)."""
    if padding_type is None:
        return None
    if not isinstance(padding_type, str):
        raise ValueError(f"parameter type of padding_type should be 'str', but got {type(padding_type)}.")
    if padding_type not in PADDINGS:
        raise NotImplementedError(f"padding type '{padding_type}' is not supported at the moment")
    if padding_type == "replicate":
        raise NotImplementedError("padding type 'replicate' is unused on the hot path and has no HIP kernel")
    return padding_type


def get_scheduler(optimizer, args, cur_it=-1):
    """reference functions.py:60-70"""
    if args.lr_policy == "lambda":
        def lambda_rule(it):
            return 1.0 - max(0, it - args.n_iter_decay) / float(args.n_iters - args.n_iter_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule, last_epoch=cur_it)
    if args.lr_policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=args.n_iter_decay, gamma=0.1, last_epoch=cur_it)
    raise NotImplementedError(f"Learning rate policy {args.lr_policy} is not implemented")


def init_weights(net, init_type="normal", init_gain=0.02):
    """Same rule as the reference (functions.py:72-94): modules whose class name STARTS with 'Conv'
    and own a weight get N(0, gain) (or xavier/kaiming/orthogonal), bias 0; nn.Linear keeps its
    default init."""
    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, "weight") and classname.find("Conv") == 0:
            # a spectrally normalised holder keeps its trainable tensor in weight_orig (the reference initialises
            # the plain ``weight`` attribute, which only aliases weight_orig while the module is still on the CPU)
            w = m.weight_orig.data if hasattr(m, "weight_orig") else m.weight.data
            if init_type == "normal":
                init.normal_(w, 0.0, init_gain)
            elif init_type == "xavier":
                init.xavier_normal_(w, gain=init_gain)
            elif init_type == "kaiming":
                init.kaiming_normal_(w, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(w, gain=init_gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if hasattr(m, "bias") and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
    print("initialize network with %s" % init_type)
    net.apply(init_func)


def init_net(net, init_type="normal", init_gain=0.02, device="cpu", gpu_ids=()):
    """Move to the device and initialise (reference functions.py:96-106).  ``gpu_ids`` is accepted for CLI
    compatibility; replicas are separate processes (torchrun), never nn.DataParallel."""
    net = net.to(device)
    if init_type:
        init_weights(net, init_type, init_gain=init_gain)
    return net


def spectral_norm(module, name="weight", n_power_iterations=1, eps=1e-12, dim=None):
    """Reference functions.py:113-121.  ``torch.nn.utils.spectral_norm`` is used for what it registers on the
    parameter holder -- ``weight_orig`` (Parameter), ``weight_u`` / ``weight_v`` (buffers) and its state-dict hooks,
    so checkpoints interchange with the reference's -- its forward pre-hook never fires because the holder's own
    forward is never called: ``conv_weight`` below computes the normalised weight with the HIP kernels."""
    if dim is None:
        dim = 1 if isinstance(module, (torch.nn.ConvTranspose1d, torch.nn.ConvTranspose2d, torch.nn.ConvTranspose3d)) else 0
    if dim != 0:
        raise NotImplementedError("spectral norm on transposed convolutions is unused by the reference models")
    module = torch.nn.utils.spectral_norm(module, name, n_power_iterations, eps, dim)
    module._mt_sn = (int(n_power_iterations), float(eps))
    return module


def conv_weight(conv, training=True):
    """The weight a conv holder contributes to the forward pass: ``conv.weight``, or for a spectrally normalised
    holder weight_orig / sigma after this call's power iteration (buffers updated in place, as torch does)."""
    sn = getattr(conv, "_mt_sn", None)
    if sn is None:
        return conv.weight
    from ... import hip_ops as ops
    return ops.spectral_norm_weight(conv.weight_orig, conv.weight_u, conv.weight_v, training=training,
                                    n_power_iterations=sn[0], eps=sn[1])
