"""GANLoss (reference src/models/core/loss.py:35-64) on the HIP loss kernels.  The VGG perceptual loss
(loss.py:66-133) needs downloaded weights and is out of scope."""
import torch.nn as nn

from ... import hip_ops as ops


class GANLoss(nn.Module):
    """'vanilla' (BCE with logits) and 'lsgan' (MSE) against constant ones / zeros; 'wgangp' = -mean(x) for a real
    target, +mean(x) for a fake one (loss.py:53-57; the reference has no gradient penalty term).  With 'hinge' the reference
    never calls this module (its ``self.loss`` is None, loss.py:46-47): the hinge terms are written out in the model
    (adain_model.py:209-210, 293-295) -- ``hinge_dis`` / ``hinge_gen`` below are those expressions."""

    def __init__(self, loss="vanilla"):
        super().__init__()
        if loss not in ("vanilla", "lsgan", "hinge", "wgangp"):
            raise NotImplementedError(f"gan_mode '{loss}' is not implemented in this build ('bce' is nn.BCELoss on raw "
                                      "logits in the reference, which torch rejects outside [0, 1])")
        self.loss_type = loss

    def forward(self, inp, trg_is_real, is_dis=None):
        if self.loss_type == "vanilla":
            return ops.bce_logits_const(inp, bool(trg_is_real))
        if self.loss_type == "lsgan":
            return ops.mse_const(inp, bool(trg_is_real))
        if self.loss_type == "wgangp":
            return ops.signed_mean(inp, negative=bool(trg_is_real))
        raise TypeError("'NoneType' object is not callable")      # what the reference raises for 'hinge' here

    @staticmethod
    def hinge_dis(pred_real, pred_fake):
        return ops.hinge_dis(pred_real, True) + ops.hinge_dis(pred_fake, False)

    @staticmethod
    def hinge_gen(pred_fake):
        return ops.neg_mean(pred_fake)


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
