"""ORACLE -- test infrastructure, not product code (see oracle/nets.py header).

CPU restatement of one training iteration of the reference's ``AdaINModel`` / ``BaseModel``
(``optimize_parameters``): discriminator update (two backward phases) followed by the generator
update (two backward phases), seven Adam steps.  Reference: src/models/adain_model.py:136-430,
src/models/base_model.py (same step logic, different networks), src/models/model.py:56-68.

Random draws are taken from an ``rng`` object in the reference's draw order (SURVEY.md
Appendix C) so a recorded reference run can be replayed exactly.
"""
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from . import nets


class TorchRng:
    """Draws with the same torch calls, in the same order, as the reference does on CPU."""

    def noise(self, shape):      # misc.py:25  torch.randn(x.size())
        return torch.randn(shape)

    def eps(self, shape):        # networks.py:132  torch.FloatTensor(size).normal_()
        return torch.FloatTensor(torch.Size(shape)).normal_()

    def z(self, shape):          # adain_model.py:84  torch.randn(bs, latent_dim)
        return torch.randn(shape)

    def mask(self, shape):       # nn.Dropout(0.5): keep-mask ~ Bernoulli(0.5)
        return torch.bernoulli(torch.full(tuple(shape), 0.5))


class ReplayRng:
    """Replays the list of tensors recorded from a reference run (tests/golden)."""

    def __init__(self, tensors, dtype=torch.float32):
        self.t = [torch.as_tensor(x).to(dtype) for x in tensors]
        self.i = 0

    def _next(self, shape):
        x = self.t[self.i]
        self.i += 1
        assert tuple(x.shape) == tuple(shape), (self.i, tuple(x.shape), tuple(shape))
        return x

    noise = eps = z = mask = _next


class RecordingRng(TorchRng):
    def __init__(self):
        self.log = []

    def _rec(self, x):
        self.log.append(x.clone())
        return x

    def noise(self, shape):
        return self._rec(super().noise(shape))

    def eps(self, shape):
        return self._rec(super().eps(shape))

    def z(self, shape):
        return self._rec(super().z(shape))

    def mask(self, shape):
        return self._rec(super().mask(shape))


def default_args(**kw):
    """Namespace with the reference's flag defaults (arguments.py:18-51,85-118)."""
    a = dict(model="AdaINModel", input_dim=3, dim=64, num_domains=2, latent_dim=8, batch_size=1, crop_size=256,
             ms_dis=False, num_scales=3, use_dis_content=False, concat=False, reparam=False,
             lr=1e-4, wd=1e-4, beta1=0.5, beta2=0.999, n_iter_decay=600000, d_iter=3,
             lambda_rec=10.0, lambda_cls=1.0, lambda_cls_G=5.0, gan_mode="vanilla", use_ragan=False, use_dropout=False, enc_norm="instance", dec_norm="layer",
             dis_norm=None)
    a.update(kw)
    return SimpleNamespace(**a)


def gan_loss(mode, x, real):
    # loss.py:52-64
    if mode == "vanilla":
        return F.binary_cross_entropy_with_logits(x, torch.ones_like(x) if real else torch.zeros_like(x))
    if mode == "lsgan":
        return F.mse_loss(x, torch.ones_like(x) if real else torch.zeros_like(x))
    if mode == "wgangp":                                     # loss.py:53-57 (no gradient penalty in the reference)
        return -x.mean() if real else x.mean()
    raise NotImplementedError(mode)


def hinge_dis(pr, pf):
    # adain_model.py:209-210
    return F.relu(1.0 - pr).mean() + F.relu(1.0 + pf).mean()


def _is_buffer(key):
    return key.endswith((".weight_u", ".weight_v", ".running_mean", ".running_var", ".num_batches_tracked"))


class OracleModel:
    """Functional twin of the reference ``Model``: ``P[net][key]`` leaf tensors + one Adam each."""

    def __init__(self, params, args, dtype=torch.float32):
        """dtype=torch.float64 gives a high-precision ground truth for gradient comparisons."""
        self.args = args
        self.dtype = dtype
        self.kind = "adain" if args.model == "AdaINModel" else "base"
        # spectral-norm vectors (weight_u / weight_v) are buffers: no gradient, not optimised
        self.P = {net: {k: torch.as_tensor(v).clone().to(dtype).requires_grad_(not _is_buffer(k))
                        for k, v in sd.items()} for net, sd in params.items()}
        self.opt, self.sched = {}, {}
        for net, sd in self.P.items():
            lr = args.lr / 2.5 if net == "content_discriminator" else args.lr      # adain_model.py:65
            self.opt[net] = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=lr,
                                             betas=(args.beta1, float(args.beta2)), weight_decay=args.wd)
            self.sched[net] = torch.optim.lr_scheduler.StepLR(self.opt[net], step_size=args.n_iter_decay, gamma=0.1)
        self.loss = {}
        self.reparam = self.kind == "adain" or args.reparam

    # ---- network dispatch ----
    def Ec(self, x, rng):
        h = nets.content_encoder(self.P["content_encoder"], x, None, getattr(self.args, "enc_norm", "instance"))
        return h + rng.noise(h.shape) if rng is not None else h

    def Es(self, x, c, rng):
        P = self.P["style_encoder"]
        if self.reparam:
            z, mu, logvar = nets.style_encoder_reparam(P, x, c, None)
            z = rng.eps(mu.shape) * torch.exp(0.5 * logvar) + mu
            return z, mu, logvar
        return nets.style_encoder_plain(P, x, c), None, None

    def Dec(self, x, z, c):
        P = self.P["decoder"]
        drop = nets._identity
        if getattr(self.args, "use_dropout", False):                     # training mode: nn.Dropout(0.5) is live
            rng = self._rng
            drop = lambda t: t * rng.mask(t.shape).to(t.dtype) / 0.5     # noqa: E731
        if self.kind == "adain":
            return nets.adain_decoder(P, x, z, c, drop)
        return nets.decoder_concat(P, x, z, c, drop) if self.args.concat else nets.decoder_plain(P, x, z, c, drop)

    def D(self, which, x):
        P = self.P[which]
        if self.args.ms_dis:
            return nets.multi_scale_discriminator(P, x, self.args.num_scales)
        return [nets.discriminator(P, x, getattr(self.args, "dis_norm", None))]

    # ---- step ----
    def update_lr(self):                                    # model.py:66-68
        for s in self.sched.values():
            s.step()

    def set_inputs(self, batch):                            # adain_model.py:87-94
        self.img = torch.cat((batch["x1"], batch["x2"]), dim=0).to(self.dtype)
        self.c_org = torch.cat((batch["y1"], batch["y2"]), dim=0).to(self.dtype)

    def _d_phase(self, which, real, fake, c_org):           # adain_model.py:182-223
        a = self.args
        self.opt[which].zero_grad()
        adv, cls = 0, 0
        hinge = "hinge" in a.gan_mode and not a.ms_dis
        ragan = a.use_ragan and not a.ms_dis                            # adain_model.py:206-208
        for (pf, _), (pr, cr) in zip(self.D(which, fake.detach()), self.D(which, real)):
            if ragan:
                adv = adv + (gan_loss(a.gan_mode, pr - torch.mean(pf), True) +
                             gan_loss(a.gan_mode, pf - torch.mean(pr), False)) / 2
            else:
                adv = adv + (hinge_dis(pr, pf) if hinge else
                             gan_loss(a.gan_mode, pf, False) + gan_loss(a.gan_mode, pr, True))
            cls = cls + F.binary_cross_entropy_with_logits(cr, c_org)
        total = adv + a.lambda_cls * cls
        total.backward()
        self.loss.update(d_adv=adv.item(), d_cls=cls.item(), d_total=total.item())
        self.opt[which].step()

    def update_discriminator(self, rng):                    # adain_model.py:146-180
        B, img, c = self.args.batch_size, self.img, self.c_org
        cls_a, cls_b = torch.split(c, B)
        z_ca, z_cb = torch.split(self.Ec(img, rng), B)
        z_s, _, _ = self.Es(img, c, rng)
        z_sa, z_sb = torch.split(z_s, B)
        z_sr = rng.z((B, self.args.latent_dim))
        img_ba, img_br = torch.split(self.Dec(torch.cat((z_cb, z_cb)), torch.cat((z_sa, z_sr)),
                                              torch.cat((cls_a, cls_a))), B)
        img_ab, img_ar = torch.split(self.Dec(torch.cat((z_ca, z_ca)), torch.cat((z_sb, z_sr)),
                                              torch.cat((cls_b, cls_b))), B)
        self._d_phase("discriminator1", img, torch.cat((img_ba, img_ab)), c)
        self._d_phase("discriminator2", img, torch.cat((img_br, img_ar)), c)

    def _g_adv(self, which, fake, c_org, real_from=None):   # adain_model.py:278-301, 352-373
        a = self.args
        adv, cls = 0, 0
        hinge = "hinge" in a.gan_mode and not a.ms_dis                  # adain_model.py:293-295, 367-369
        if a.use_ragan and not a.ms_dis:                                # 286-292, 360-366 (relativistic average)
            (pr, _), = self.D(real_from, self.img)
            (pf, cf), = self.D(which, fake)
            adv = (gan_loss(a.gan_mode, pr - torch.mean(pf), False) + gan_loss(a.gan_mode, pf - torch.mean(pr), True)) / 2
            return adv, F.binary_cross_entropy_with_logits(cf, c_org) * a.lambda_cls_G
        for pf, cf in self.D(which, fake):
            adv = adv + (-pf.mean() if hinge else gan_loss(a.gan_mode, pf, True))
            cls = cls + F.binary_cross_entropy_with_logits(cf, c_org)
        return adv, cls * a.lambda_cls_G

    def backward_generator(self, rng):                      # adain_model.py:241-332
        a, B, img, c = self.args, self.args.batch_size, self.img, self.c_org
        cls_a, cls_b = torch.split(c, B)
        z_c = self.Ec(img, rng)
        z_ca, z_cb = torch.split(z_c, B)
        z_s, mu, logvar = self.Es(img, c, rng)
        z_sa, z_sb = torch.split(z_s, B)
        img_ba, img_aa = torch.split(self.Dec(torch.cat((z_cb, z_ca)), torch.cat((z_sa, z_sa)),
                                              torch.cat((cls_a, cls_a))), B)
        img_ab, img_bb = torch.split(self.Dec(torch.cat((z_ca, z_cb)), torch.cat((z_sb, z_sb)),
                                              torch.cat((cls_b, cls_b))), B)
        img_fake = torch.cat((img_ba, img_ab))
        img_self = torch.cat((img_aa, img_bb))
        z_c_rec_b, z_c_rec_a = torch.split(self.Ec(img_fake, rng), B)          # note the swapped order (264-265)
        z_s_rec, _, _ = self.Es(img_fake, c, rng)
        z_s_rec_a, z_s_rec_b = torch.split(z_s_rec, B)
        img_recon = self.Dec(torch.cat((z_c_rec_a, z_c_rec_b)), torch.cat((z_s_rec_a, z_s_rec_b)),
                             torch.cat((cls_a, cls_b)))
        g_content = None
        if a.use_dis_content:
            pred = nets.content_discriminator(self.P["content_discriminator"], z_c)
            g_content = F.binary_cross_entropy_with_logits(pred, 1 - c)
        g_adv, g_cls = self._g_adv("discriminator1", img_fake, c, real_from="discriminator1")
        l_self = F.l1_loss(img, img_self) * a.lambda_rec
        l_cc = F.l1_loss(img, img_recon) * a.lambda_rec
        kl_zc = torch.mean(z_c ** 2) * 0.01
        if self.reparam:
            kl_zs = torch.sum(1 + logvar - mu ** 2 - logvar.exp()) * -0.5 * 0.01   # a SUM (313-314)
        else:
            kl_zs = torch.mean(z_s ** 2) * 0.01
        total = g_adv + g_cls + l_self + l_cc + kl_zc + kl_zs
        if g_content is not None:
            total = total + g_content
            self.loss["g_content"] = g_content.item()
        total.backward()
        self.loss.update(g_adv=g_adv.item(), g_cls=g_cls.item(), kl_zc=kl_zc.item(), kl_zs=kl_zs.item(),
                         l1_self_rec=l_self.item(), l1_cc_rec=l_cc.item(), total_g=total.item())
        self.fakes = dict(img_fake=img_fake.detach(), img_self=img_self.detach(), img_recon=img_recon.detach())

    def backward_decoder_random(self, rng):                 # adain_model.py:339-394
        a, B, img, c = self.args, self.args.batch_size, self.img, self.c_org
        cls_a, cls_b = torch.split(c, B)
        z_ca, z_cb = torch.split(self.Ec(img, rng), B)
        z_sr = rng.z((B, a.latent_dim))
        img_random = torch.cat((self.Dec(z_cb, z_sr, cls_a), self.Dec(z_ca, z_sr, cls_b)))
        # --ms_dis scores with discriminator1 here (reference quirk, lines 352-353)
        # ... and with --use_ragan the fake logits come from discriminator1, the real ones from discriminator2 (360-362)
        ragan = a.use_ragan and not a.ms_dis
        adv, cls = self._g_adv("discriminator1" if (a.ms_dis or ragan) else "discriminator2", img_random, c,
                               real_from="discriminator2")
        if self.reparam:
            _, mu2, _ = self.Es(img_random, c, rng)
            m_a, m_b = torch.split(mu2, B)
            lz = (F.l1_loss(m_a, z_sr) + F.l1_loss(m_b, z_sr)) * 10
        else:
            zr, _, _ = self.Es(img_random, c, rng)
            zr_a, _ = torch.split(zr, B)
            lz = (F.l1_loss(zr_a, z_sr) + F.l1_loss(zr_a, z_sr)) * 10          # base_model.py:419-420 (a twice)
        total = lz + adv + cls
        total.backward()
        self.loss.update(l1_recon_z=lz.item(), gan2=adv.item(), gan2_cls=cls.item())
        self.fakes["img_random"] = img_random.detach()

    def update_generator(self, rng):                        # adain_model.py:225-239
        for n in ("content_encoder", "style_encoder", "decoder"):
            self.opt[n].zero_grad()
        self.backward_generator(rng)
        for n in ("content_encoder", "style_encoder", "decoder"):
            self.opt[n].step()
        for n in ("content_encoder", "decoder"):
            self.opt[n].zero_grad()
        self.backward_decoder_random(rng)
        for n in ("content_encoder", "decoder"):
            self.opt[n].step()

    def update_content_discriminator(self, rng):            # adain_model.py:136-144
        z_c = self.Ec(self.img, rng)
        self.opt["content_discriminator"].zero_grad()
        pred = nets.content_discriminator(self.P["content_discriminator"], z_c.detach())
        loss = F.binary_cross_entropy_with_logits(pred, self.c_org)
        loss.backward()
        self.loss_dc = loss.item()
        torch.nn.utils.clip_grad_norm_(list(self.P["content_discriminator"].values()), 5)
        self.opt["content_discriminator"].step()

    def optimize_parameters(self, it, rng=None):            # adain_model.py:421-430
        rng = rng or TorchRng()
        self._rng = rng                                         # (Dec draws its dropout masks from the same stream)
        if self.args.use_dis_content and it % self.args.d_iter != 0:
            self.update_content_discriminator(rng)
            return
        self.update_discriminator(rng)
        self.update_generator(rng)

    # ---- inference (sample.py:79-91 -> adain_model.py:96-109) ----
    # sample.py never calls .eval(): the content encoder's GaussianNoiseLayer stays ACTIVE while sampling (SURVEY Appendix
    # D-12), so both calls draw one noise tensor (rng=None: the noise-free eval behaviour)
    @torch.no_grad()
    def forward_random(self, img, z_r, c_trg, rng=None):     # adain_model.py:96-101
        z_c = self.Ec(img.to(self.dtype), rng)
        return self.Dec(z_c, z_r.to(self.dtype), c_trg.to(self.dtype))

    @torch.no_grad()
    def forward_reference(self, img_src, img_ref, c_trg, rng=None):   # adain_model.py:103-109
        z_c = self.Ec(img_src.to(self.dtype), rng)
        c = c_trg.to(self.dtype)
        z_s, _, _ = self.Es(img_ref.to(self.dtype), c, rng if rng is not None else _ZeroEps())
        return self.Dec(z_c, z_s, c)

    def state(self):
        return {net: {k: v.detach().clone() for k, v in sd.items()} for net, sd in self.P.items()}


class _ZeroEps:
    """eval-mode reparameterisation: z = mu"""
    def eps(self, shape):
        return torch.zeros(tuple(shape))
