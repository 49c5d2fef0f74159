"""GANLoss (reference src/models/core/loss.py:35-64) on the HIP loss kernels.  The VGG perceptual loss
(loss.py:66-133) needs downloaded weights and is out of scope."""
import torch.nn as nn

from ... import hip_ops as ops


class GANLoss(nn.Module):
    def __init__(self, loss="vanilla"):
        super().__init__()
        if loss != "vanilla":
            raise NotImplementedError(f"gan_mode '{loss}' has no HIP kernel in this build (only 'vanilla', the "
                                      "reference default, is on the north-star path)")
        self.loss_type = loss

    def forward(self, inp, trg_is_real, is_dis=None):
        return ops.bce_logits_const(inp, bool(trg_is_real))


class ClassificationLoss(nn.Module):
    """nn.BCEWithLogitsLoss() on [N, D] class logits (adain_model.py:74)."""

    def forward(self, logits, target):
        return ops.bce_logits(logits, target)


class L1Loss(nn.Module):
    """nn.L1Loss() (mean)"""

    def forward(self, a, b):
        return ops.l1_loss(a, b)


class VGGPerceptualLoss(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("--vgg_loss needs torchvision's pretrained VGG (a download); out of scope")
