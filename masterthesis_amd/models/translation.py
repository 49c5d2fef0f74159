"""The G+D training step shared by ``AdaINModel`` and ``BaseModel``.

Reference: src/models/adain_model.py:83-430 and src/models/base_model.py (identical step logic around
different generator networks).  Same four backward phases and seven Adam steps:

    PHASE 1  D1 on (img, img_fake)      PHASE 2  D2 on (img, img_random)          (update_discriminator)
    PHASE 3  Ec, Es, Dec  (translation, self/cross-cycle reconstruction, KL)        (backward_generator)
    PHASE 4  Ec, Dec      (random-style translation, latent regression)             (backward_decoder_random)

MI355X-first execution choices (all trajectory-identical to the reference):
 * the generator forwards that feed the discriminator update run under ``no_grad`` (the reference builds
   an autograd graph there and throws it away);
 * discriminator weight gradients are not computed in the generator phases (the reference computes
   them and discards them at the next ``zero_grad``);
 * loss scalars stay on the device; ``.item()``-style host syncs (16 per step in the reference) happen
   only when the train loop prints / logs;
 * after each backward the flat gradient buffers of the networks about to be stepped are all-reduced
   (mean) over RCCL on a side stream; the KL term is a batch SUM in the reference
   (adain_model.py:313-314) so its gradient is scaled by world_size to stay equal to one process at
   the global batch.
"""
import os
import time

import torch

from .. import hip_ops as ops
from ..optim import FusedAdam
from .core import loss as losses
from .core import networks
from .core.misc import random_source
from .model import Model


class TranslationModel(Model):
    """Common constructor tail + step.  Subclasses build content_encoder / style_encoder / decoder."""

    reparam = True

    def _build_training_side(self, args):
        if "train" not in args.mode:
            return
        if args.ms_dis:
            def make_d():
                # the reference never passes ``dim`` (adain_model.py:33-42): width 64 unless --ms_dim says otherwise
                return networks.MultiScaleDiscriminator(args.input_dim, dim=getattr(args, "ms_dim", None) or 64,
                                                        norm_layer=args.dis_norm, sn=args.dis_sn,
                                                        num_domains=args.num_domains, num_scales=args.num_scales)
        else:
            def make_d():
                return networks.Discriminator(args.input_dim, dim=args.dim, norm_layer=args.dis_norm, sn=args.dis_sn,
                                              num_domains=args.num_domains, image_size=args.crop_size)
        self.model.discriminator1 = make_d()
        self.model.discriminator2 = make_d()
        for net in self.model:
            self.optimizer[net] = FusedAdam(self.model[net].parameters(), lr=args.lr,
                                            betas=(args.beta1, float(args.beta2)), weight_decay=args.wd)
        if args.use_dis_content:
            self.model.content_discriminator = networks.ContentDiscriminator(
                dim=self.model.content_encoder.output_dim, num_domains=args.num_domains)
            self.optimizer.content_discriminator = FusedAdam(
                self.model.content_discriminator.parameters(), lr=args.lr / 2.5,
                betas=(args.beta1, float(args.beta2)), weight_decay=args.wd)
        self.gan_loss = losses.GANLoss(args.gan_mode)
        self.classification_loss = losses.ClassificationLoss()
        self.l1_loss = losses.L1Loss()
        if args.vgg_loss is not None:
            raise NotImplementedError("--vgg_loss needs downloaded VGG weights; out of scope")
        self.print_loss = ["g_adv", "g_cls", "l1_cc_rec"]
        # conv weight gradients are accumulated straight into FusedAdam's flat gradient views
        ops.set_fused_grad_accumulation(True)

    # ---- small helpers -------------------------------------------------------------------------
    def get_z_random(self, bs, latent_dim):
        return random_source().z((bs, latent_dim), self.device)

    def set_inputs(self, inputs):
        self.img_a = inputs["x1"].to(self.device, non_blocking=True).detach()
        self.cls_a = inputs["y1"].to(self.device, non_blocking=True).detach().float()
        self.img_b = inputs["x2"].to(self.device, non_blocking=True).detach()
        self.cls_b = inputs["y2"].to(self.device, non_blocking=True).detach().float()
        img = ops.canon(torch.cat((self.img_a, self.img_b), dim=0))     # one NCHW -> padded-NHWC pass
        c_org = torch.cat((self.cls_a, self.cls_b), dim=0)
        if self._graph_mode():
            # a captured step reads its batch from fixed addresses: copy into the persistent input buffers
            st = self.__dict__.get("_static_in")
            if st is None or st[0].shape != img.shape or st[0].dtype != img.dtype or st[1].shape != c_org.shape:
                self._static_in = (img, c_org)
                self._graphs = {}
            else:
                st[0].copy_(img)
                st[1].copy_(c_org)
            img, c_org = self._static_in
        self.img, self.c_org = img, c_org

    # ---- hipGraph execution of the step (--hip_graph / MT_GRAPH=1) ------------------------------------------
    GRAPH_WARMUP = 3

    def _graph_mode(self):
        """The whole ``optimize_parameters`` call (forward, the four backward phases, seven Adam steps, weight re-packs:
        ~1700 launches) is captured once into a hipGraph and replayed: the Python/ctypes enqueue cost (20-35 ms per step,
        DESIGN section 4) drops to one graph launch, which is what makes the reference's own ``--batch_size 1``
        (scripts/train.sh) GPU-bound.  Everything that changes between steps lives in device memory: the batch (static
        input buffers), the Philox state of the noise kernels, Adam's step count / learning rate.  Not with a live
        gradient exchange (RCCL calls stay outside graphs here) and not when random draws are injected from the host
        (parity tests)."""
        on = getattr(self.args, "hip_graph", False)
        env = os.environ.get("MT_GRAPH")
        if env is not None:
            on = env == "1"
        from .core.misc import DeviceRandom
        return (bool(on) and "train" in self.args.mode and not self.reducer.enabled
                and isinstance(random_source(), DeviceRandom) and torch.cuda.is_available())

    def _optimize_graphed(self, global_iter):
        kind = "dc" if (self.args.use_dis_content and global_iter % self.args.d_iter != 0) else "full"
        opts = list(self.optimizer.values())
        # (anything that changes the launch sequence invalidates the capture: rebuilt flat buffers, train / eval flags,
        #  deterministic mode, the storage type)
        key = (kind, tuple(o.generation for o in opts), ops.compute_dtype(), ops.deterministic(),
               tuple(self.model[n].training for n in self.model), ops.graph_epoch())
        graphs = self.__dict__.setdefault("_graphs", {})
        for k in [k for k in graphs if k[0] == kind and k != key and (k[1] != key[1] or k[5] != key[5])]:
            del graphs[k]       # captured against flat buffers / arena / pack tables that no longer exist: never replayable
        st = graphs.setdefault(key, {"warm": 0})
        if st["warm"] < self.GRAPH_WARMUP:           # eager steps first: caches, flat buffers, pack tables, arenas settle
            st["warm"] += 1
            return self._optimize_eager(global_iter)
        if "graph" not in st:
            for o in opts:
                o.sync_lr()                          # (so that no learning-rate write is captured)
            before = [o._step_count_mt for o in opts]
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._optimize_eager(global_iter)
            st["graph"] = g
            st["steps"] = [o._step_count_mt - b for o, b in zip(opts, before)]
            g.replay()                               # capture records, it does not execute: run this iteration now
            return
        for o in opts:
            o.sync_lr()
        st["graph"].replay()
        for o, n in zip(opts, st["steps"]):
            o.advance_host_step(n)

    def _encode_style(self, img, c):
        out = self.model.style_encoder(img, c)
        return out if self.reparam else (out, None, None)

    # ---- encoder passes shared by the discriminator update and phase 3 ---------------------------------
    def _share_encoders(self):
        """The reference encodes ``img`` in update_discriminator (adain_model.py:153-156) and again at the top of
        backward_generator (244-247) with the SAME encoder weights (only the discriminators step in between): up to the
        random draws -- content noise, reparameterisation eps -- the two passes compute identical numbers.  Here the
        deterministic part (ContentEncoder.features, the style encoder's moments) runs once per step, with its autograd
        graph, and each phase applies its own fresh draws (in the reference's draw order).  Not with BatchNorm in the
        encoders (every forward call updates the running statistics) and not outside training mode."""
        a = self.args
        Ec, Es = self.model.content_encoder, self.model.style_encoder
        return (Ec.training and Es.training and getattr(a, "enc_norm", "instance") != "batch"
                and hasattr(Ec, "features") and os.environ.get("MT_NO_ENCODER_SHARING", "0") != "1")

    def _encode_shared(self, img, c_org):
        Ec, Es = self.model.content_encoder, self.model.style_encoder
        h = Ec.features(img)
        if self.reparam:
            return h, Es.moments(img, c_org)
        return h, (Es(img, c_org),)

    def _draw_from_shared(self, shared, detach):
        """(z_c, z_s, mu, logvar) from the shared deterministic part with fresh draws"""
        h, st = shared
        if detach:
            h, st = h.detach(), tuple(t.detach() for t in st)
        z_c = self.model.content_encoder.add_noise(h)
        if self.reparam:
            mu, logvar = st
            return z_c, self.model.style_encoder.reparameterize(mu, logvar), mu, logvar
        return z_c, st[0], None, None

    def _translate(self, contents, styles, classes, per_call=None):
        """Several translations in one decoder call.  ``per_call``: how many of the parts the reference feeds to ONE
        decoder call -- with --use_dropout the calls are kept apart so the masks are drawn in the reference's order, with
        --dec_norm batch because the batch statistics are those of one call."""
        B = self.args.batch_size
        apart = getattr(self.args, "use_dropout", False) or getattr(self.args, "dec_norm", None) == "batch"
        if per_call and apart and self.model.decoder.training:
            outs = []
            for i in range(0, len(contents), per_call):
                sl = slice(i, i + per_call)
                outs += list(self._translate(contents[sl], styles[sl], classes[sl]))
            return tuple(outs)
        fake = self.model.decoder(ops.cat_batch(contents), torch.cat(styles, dim=0), torch.cat(classes, dim=0))
        return torch.split(fake, B, dim=0)

    def _reduce_and_step(self, names, early=None):
        """exchange (what the backward pass has not already sent) + Adam step of every network of a phase"""
        self._mark("exchange " + "+".join(names))
        opts = [self.optimizer[n] for n in names]
        if early is None:
            handles = [[h] for h in self.reducer.reduce([o.flat_grad() for o in opts])]
        else:
            handles = [self._finish_early_exchange(n, early[n]) for n in names]
        for o, hs in zip(opts, handles):
            for h in hs:
                self.reducer.wait(h)
            o.step()

    # ---- inference surface (reference adain_model.py:96-134) --------------------------------------
    def _infer(self, name, fn, *inputs):
        """Run an inference forward; with --hip_graph (and no autograd graph wanted) replay it from a hipGraph captured
        per input signature after two eager calls: at batch 1 the ~120 launches of a 540x960 translation are launch
        bound.  The returned tensor is the graph's static output buffer: valid until the next call of the same kind
        (sample.py converts it to NCHW fp32 right away)."""
        from .core.misc import DeviceRandom
        on = getattr(self.args, "hip_graph", False)
        env = os.environ.get("MT_GRAPH")
        if env is not None:
            on = env == "1"
        if not (on and not torch.is_grad_enabled() and torch.cuda.is_available()
                and isinstance(random_source(), DeviceRandom)):
            return fn(*inputs)
        # (the weights' versions: a checkpoint load or an optimizer step re-packs the weight images outside the graph)
        wver = tuple(sum(p._version for p in self.model[n].parameters()) for n in self.model)
        key = (name, ops.compute_dtype(), tuple((tuple(t.shape), t.dtype) for t in inputs),
               tuple(self.model[n].training for n in self.model), ops.graph_epoch(), wver)
        graphs = self.__dict__.setdefault("_infer_graphs", {})
        for k in [k for k in graphs if k[0] == name and k != key and k[4:] != key[4:]]:
            del graphs[k]       # captured against other weights / freed scratch: never replayable
        st = graphs.setdefault(key, {"warm": 0})
        if st["warm"] < 2:
            st["warm"] += 1
            return fn(*inputs)
        if "graph" not in st:
            st["in"] = [t.detach().clone() for t in inputs]
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                st["out"] = fn(*st["in"])
            st["graph"] = g
        for s_, t in zip(st["in"], inputs):
            s_.copy_(t)
        st["graph"].replay()
        return st["out"]

    def forward_random(self, img, z_r, c_trg):
        start = time.time()
        img_fake = self._infer("random", lambda i, z, c: self.model.decoder(self.model.content_encoder(i), z, c),
                               img, z_r, c_trg)
        end = time.time()
        return img_fake, end - start, torch.cuda.memory_reserved(0) / (1024 * 1024 * 1024)

    def forward_reference(self, img_src, img_ref, c_trg):
        start = time.time()

        def fn(src, ref, c):
            z_c = self.model.content_encoder(src)
            z_s, _, _ = self._encode_style(ref, c)
            return self.model.decoder(z_c, z_s, c)
        img_fake = self._infer("reference", fn, img_src, img_ref, c_trg)
        end = time.time()
        return img_fake, end - start, torch.cuda.memory_reserved(0) / (1024 * 1024 * 1024)

    def forward(self, img, c_org):
        B = self.args.batch_size
        z_ca, z_cb = torch.split(self.model.content_encoder(img), B, dim=0)
        z_s, _, _ = self._encode_style(img, c_org)
        z_sa, z_sb = torch.split(z_s, B, dim=0)
        z_sr = self.get_z_random(B, self.args.latent_dim)
        cls_a, cls_b = torch.split(c_org, B, dim=0)
        img_ba, img_aa, img_br = self._translate((z_cb, z_ca, z_cb), (z_sa, z_sa, z_sr), (cls_a, cls_a, cls_a))
        img_ab, img_bb, img_ar = self._translate((z_ca, z_cb, z_ca), (z_sb, z_sb, z_sr), (cls_b, cls_b, cls_b))
        return (ops.cat_batch((img_ba, img_ab)), ops.cat_batch((img_br, img_ar)), ops.cat_batch((img_aa, img_bb)))

    # ---- content discriminator (optional; reference adain_model.py:136-144, 334-337) ---------------
    @ops.step_scope
    def update_content_discriminator(self, img, c_org):
        with torch.no_grad():
            z_c = self.model.content_encoder(img)
        self.optimizer.content_discriminator.zero_grad()
        pred = self.model.content_discriminator(z_c.detach())
        loss_d_content = self.classification_loss(pred, c_org)
        loss_d_content.backward()
        self.loss_dc = loss_d_content.detach()
        opt = self.optimizer.content_discriminator
        (h,) = self.reducer.reduce([opt.flat_grad()])
        self.reducer.wait(h)
        g = opt.flat_grad()
        g.mul_(torch.clamp(5.0 / (g.norm() + 1e-6), max=1.0))      # clip_grad_norm_(…, 5)
        opt.step()

    def backward_content_discriminator(self, z_c):
        pred = self.model.content_discriminator(z_c)
        return self.classification_loss(pred, 1 - self.c_org)

    # ---- PHASE 1-2 ----------------------------------------------------------------------------------
    @ops.step_scope
    def update_discriminator(self, img, c_org):
        self._finish_deferred()
        B = self.args.batch_size
        cls_a, cls_b = torch.split(c_org, B, dim=0)
        self._shared = self._encode_shared(img, c_org) if self._share_encoders() else None
        with torch.no_grad():
            if self._shared is not None:
                z_c, z_s, _, _ = self._draw_from_shared(self._shared, detach=True)
            else:
                z_c = self.model.content_encoder(img)
                z_s, _, _ = self._encode_style(img, c_org)
            z_ca, z_cb = torch.split(z_c, B, dim=0)
            z_sa, z_sb = torch.split(z_s, B, dim=0)
            z_sr = self.get_z_random(B, self.args.latent_dim)
            # all four translations in ONE decoder call (per-sample AdaIN / LayerNorm: identical outputs, half
            # the launches, 4B images per GEMM)
            img_ba, img_br, img_ab, img_ar = self._translate((z_cb, z_cb, z_ca, z_ca), (z_sa, z_sr, z_sb, z_sr),
                                                             (cls_a, cls_a, cls_b, cls_b), per_call=2)
            img_fake = ops.cat_batch((img_ba, img_ab))
            img_random = ops.cat_batch((img_br, img_ar))
        # the gradient exchange of discriminator1 (side stream) overlaps discriminator2's forward + backward
        pending = []
        for name, fake in (("discriminator1", img_fake), ("discriminator2", img_random)):
            self._mark(name)
            opt = self.optimizer[name]
            opt.zero_grad()
            early = self._arm_early_exchange(name, nbuckets=2)
            self.backward_discriminator(self.model[name], img, fake, c_org)
            self._mark("backward done " + name)
            pending.append((opt, self._finish_early_exchange(name, early)))
        # discriminator1 is needed by phase 3; discriminator2 only by phase 4, so with a live exchange its wait + Adam
        # step move in front of phase 4 and its all-reduce also hides behind the whole of phase 3
        (opt1, h1), d2 = pending
        for h in h1:
            self.reducer.wait(h)
        opt1.step()
        if self.reducer.enabled:
            self._deferred_steps = [d2]
        else:
            self._finish_deferred([d2])

    def _finish_deferred(self, items=None):
        items = self.__dict__.pop("_deferred_steps", []) if items is None else items
        for opt, handles in items:
            for h in handles:
                self.reducer.wait(h)
            opt.step()

    def _mark(self, what):
        if self.reducer.log is not None:
            self.reducer.log.append(("phase", what))

    # ---- exchange that starts inside the backward pass ----------------------------------------------------
    def _gradient_sources(self, name):
        """How each parameter of network ``name`` receives its gradient: "kernel" = accumulated straight into the flat
        buffer by our backward kernels (convolution / linear weights; their biases ride on the same launch), "autograd" =
        through autograd's AccumulateGrad (LayerNorm / BatchNorm affine parameters).  None if the network has parameters
        we cannot track (spectral-norm holders: the kernels see weight_orig / sigma, a non-leaf)."""
        cache = self.__dict__.setdefault("_grad_sources", {})
        if name not in cache:
            import torch.nn as nn
            kernel, rides, auto, ok = [], set(), [], True
            for m in self.model[name].modules():
                if hasattr(m, "weight_orig"):
                    ok = False
                own = dict(m.named_parameters(recurse=False))
                if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d, nn.Linear)):
                    if "weight" in own:
                        kernel.append(own["weight"])
                    if "bias" in own:
                        rides.add(id(own["bias"]))
                else:
                    auto += list(own.values())
            cache[name] = (kernel, rides, auto) if ok else None
        return cache[name]

    def _arm_early_exchange(self, name, nbuckets=2):
        """Bucketed all-reduce launched from the backward pass itself: as soon as the last gradient contribution of every
        parameter of a bucket has been enqueued, its slice of the flat gradient buffer goes to the side stream.  For a
        discriminator the 9.4 M-parameter last convolution is produced FIRST, so most of the exchange runs under the rest
        of the backward pass; in the generator phases the decoder's buffer travels under the encoders' backward.
        "Last contribution enqueued" = the per-weight use counter of hip_ops returned to zero (kernel-accumulated
        parameters) or autograd's post-accumulate hook fired (affine norm parameters)."""
        opt = self.optimizer[name]
        src = self._gradient_sources(name) if self.reducer.enabled else None
        if src is None or os.environ.get("MT_NO_EARLY_EXCHANGE", "0") == "1":
            return None
        kernel, rides, auto = src
        tracked = {id(p) for p in kernel} | {id(p) for p in auto}
        state = {"handles": [], "buckets": [], "params": []}
        flat = opt.flat_grad()
        for lo, hi, members in opt.grad_buckets(nbuckets):
            mem = [p for p in members if id(p) in tracked]
            b = {"lo": lo, "hi": hi, "left": len(mem), "sent": False}
            state["buckets"].append(b)

            def ready(_p, b=b):
                b["left"] -= 1
                if b["left"] == 0 and not b["sent"]:
                    b["sent"] = True
                    state["handles"] += self.reducer.reduce([flat[b["lo"]:b["hi"]]])
            for p in mem:
                ops.set_grad_ready_hook(p, ready)
            state["params"] += mem
        for p in auto:
            if not getattr(p, "_mt_post_hooked", False):       # registered once; dispatches to whatever is armed
                p.register_post_accumulate_grad_hook(
                    lambda q: (getattr(q, "_mt_ready_hook", None) or (lambda _q: None))(q))
                p._mt_post_hooked = True
        return state

    def _finish_early_exchange(self, name, state):
        """-> the handles the optimizer step has to wait for"""
        opt = self.optimizer[name]
        if state is None:
            return self.reducer.reduce([opt.flat_grad()])
        for p in state["params"]:
            ops.set_grad_ready_hook(p, None)
        flat = opt.flat_grad()
        for b in state["buckets"]:              # (a bucket with a parameter that was not used in this graph)
            if not b["sent"]:
                b["sent"] = True
                state["handles"] += self.reducer.reduce([flat[b["lo"]:b["hi"]]])
        return state["handles"]

    def _dis_outputs(self, netD, x):
        out = netD(x)
        return out if self.args.ms_dis else [out]

    def backward_discriminator(self, netD, real, fake, c_org):
        """covers the reference's backward_discriminator and backward_multi_scale_discriminator (182-223).
        The fake and the real batch go through the discriminator as ONE concatenated batch (same weights,
        per-sample independent layers => identical outputs, half the launches, twice the pixels per GEMM) unless
        --dis_sn makes the weights depend on the call."""
        n = fake.shape[0]
        adv_terms, cls_terms = [], []
        hinge = "hinge" in self.args.gan_mode and not self.args.ms_dis     # (adain_model.py:209; ms_dis goes via gan_loss)
        ragan = getattr(self.args, "use_ragan", False) and not self.args.ms_dis      # (adain_model.py:206-208)
        if getattr(self.args, "dis_sn", False) or getattr(self.args, "dis_norm", None) == "batch":
            # spectral norm runs one power iteration per forward call, so the weights of the two calls differ:
            # fake first, then real, as the reference does (adain_model.py:184-185, 203-205)
            outs_f = self._dis_outputs(netD, fake.detach())
            outs = [(pf, pr, cr) for (pf, _), (pr, cr) in zip(outs_f, self._dis_outputs(netD, real))]
        else:
            outs = []
            for pred, cls in self._dis_outputs(netD, ops.cat_batch((fake.detach(), real))):
                pf, pr = torch.split(pred, n, dim=0)
                outs.append((pf, pr, cls[n:]))
        for pf, pr, cr in outs:
            if ragan:
                adv_terms += [(self.gan_loss(ops.sub_mean(pr, pf), 1), 0.5), (self.gan_loss(ops.sub_mean(pf, pr), 0), 0.5)]
            elif hinge:
                adv_terms += [(ops.hinge_dis(pr, True), 1.0), (ops.hinge_dis(pf, False), 1.0)]
            else:
                adv_terms += [(self.gan_loss(pf, 0), 1.0), (self.gan_loss(pr, 1), 1.0)]
            cls_terms.append((self.classification_loss(cr, c_org), 1.0))
        # loss_d = loss_d_adv + lambda_cls * loss_d_cls: one launch for the whole expression (and one in backward)
        loss_d, v, total = ops.loss_sum([("d_adv", adv_terms, 1.0, 1.0), ("d_cls", cls_terms, self.args.lambda_cls, self.args.lambda_cls)])
        loss_d.backward()
        self._set_loss(d_adv=v["d_adv"], d_cls=v["d_cls"], d_total=total)

    backward_multi_scale_discriminator = backward_discriminator

    # ---- PHASE 3-4 ----------------------------------------------------------------------------------
    @ops.step_scope
    def update_generator(self, img, c_org):
        names = ("content_encoder", "style_encoder", "decoder")
        self._mark("phase3")
        for n in names:
            self.optimizer[n].zero_grad()
        early = {n: self._arm_early_exchange(n, nbuckets=1) for n in names}    # a network's buffer leaves when it is complete
        self.backward_generator(img, c_org)
        self._reduce_and_step(names, early)
        self._finish_deferred()                 # discriminator2's step (deferred by update_discriminator)
        names = ("content_encoder", "decoder")
        self._mark("phase4")
        for n in names:
            self.optimizer[n].zero_grad()
        early = {n: self._arm_early_exchange(n, nbuckets=1) for n in names}
        self.backward_decoder_random(img, c_org)
        self._reduce_and_step(names, early)

    def _generator_adv(self, netD, fake, c_org, netD_real=None):
        """-> ([(adversarial term, weight)], [(classification term, weight)]) of the generator loss"""
        adv, cls = [], []
        hinge = "hinge" in self.args.gan_mode and not self.args.ms_dis     # (adain_model.py:293-295, 367-369)
        lam = self.args.lambda_cls_G
        if getattr(self.args, "use_ragan", False) and not self.args.ms_dis:
            # relativistic average (286-292, 360-366): real logits from netD_real on the input images
            with ops.frozen(netD, netD_real):
                pr, _ = netD_real(self.img)
                pf, cf = netD(fake)
                adv = [(self.gan_loss(ops.sub_mean(pr, pf), 0), 0.5), (self.gan_loss(ops.sub_mean(pf, pr), 1), 0.5)]
                return adv, [(self.classification_loss(cf, c_org), lam)]
        with ops.frozen(netD):
            for pf, cf in self._dis_outputs(netD, fake):
                adv.append((self.gan_loss.hinge_gen(pf) if hinge else self.gan_loss(pf, 1), 1.0))
                cls.append((self.classification_loss(cf, c_org), lam))
        return adv, cls

    def backward_generator(self, img, c_org):
        a, B = self.args, self.args.batch_size
        cls_a, cls_b = torch.split(c_org, B, dim=0)
        shared = self.__dict__.pop("_shared", None)
        if shared is not None:
            z_c, z_s, mu, logvar = self._draw_from_shared(shared, detach=False)
            del shared
        else:
            z_c = self.model.content_encoder(img)
            z_s, mu, logvar = self._encode_style(img, c_org)
        z_ca, z_cb = torch.split(z_c, B, dim=0)
        z_sa, z_sb = torch.split(z_s, B, dim=0)
        img_ba, img_aa, img_ab, img_bb = self._translate((z_cb, z_ca, z_ca, z_cb), (z_sa, z_sa, z_sb, z_sb),
                                                         (cls_a, cls_a, cls_b, cls_b), per_call=2)    # one 4B-image decoder call
        img_fake = ops.cat_batch((img_ba, img_ab))
        img_self = ops.cat_batch((img_aa, img_bb))
        # cross-cycle: re-encode the translations (note the swapped split order, adain_model.py:264-265)
        z_c_rec_b, z_c_rec_a = torch.split(self.model.content_encoder(img_fake), B, dim=0)
        z_s_rec, _, _ = self._encode_style(img_fake, c_org)
        z_s_rec_a, z_s_rec_b = torch.split(z_s_rec, B, dim=0)
        img_recon = self.model.decoder(ops.cat_batch((z_c_rec_a, z_c_rec_b)), torch.cat((z_s_rec_a, z_s_rec_b), dim=0),
                                       torch.cat((cls_a, cls_b), dim=0))
        groups = []
        if a.use_dis_content:
            with ops.frozen(self.model.content_discriminator):
                groups.append(("g_content", [(self.backward_content_discriminator(z_c), 1.0)], 1.0, 1.0))
        adv, cls = self._generator_adv(self.model.discriminator1, img_fake, c_org, netD_real=self.model.discriminator1)
        groups += [("g_adv", adv, 1.0, 1.0), ("g_cls", cls, 1.0, 1.0),
                   ("l1_self_rec", [(self.l1_loss(img, img_self), a.lambda_rec)], 1.0, 1.0),
                   ("l1_cc_rec", [(self.l1_loss(img, img_recon), a.lambda_rec)], 1.0, 1.0),
                   ("kl_zc", [(ops.l2_mean(z_c), 0.01)], 1.0, 1.0)]
        if self.reparam:
            # a batch SUM (adain_model.py:313-314): its gradient is scaled by world_size before the mean all-reduce (see
            # the module docstring); the reported values stay those of this rank's shard
            groups.append(("kl_zs", [(ops.kl_sum(mu, logvar), 0.01)], float(self.reducer.world), 1.0))
        else:
            groups.append(("kl_zs", [((z_s * z_s).mean(), 0.01)], 1.0, 1.0))
        loss_g, v, total = ops.loss_sum(groups)
        loss_g.backward()
        self._set_loss(total_g=total, **v)

    def backward_decoder_random(self, img, c_org):
        a, B = self.args, self.args.batch_size
        cls_a, cls_b = torch.split(c_org, B, dim=0)
        z_ca, z_cb = torch.split(self.model.content_encoder(img), B, dim=0)
        z_sr = self.get_z_random(B, a.latent_dim)
        img_br, img_ar = self._translate((z_cb, z_ca), (z_sr, z_sr), (cls_a, cls_b), per_call=1)        # one 2B-image decoder call
        img_random = ops.cat_batch((img_br, img_ar))
        # with --ms_dis the reference scores the random translations with discriminator1 (352-353)
        # ... and with --use_ragan the fake logits come from discriminator1, the real ones from discriminator2 (360-362)
        ragan = getattr(a, "use_ragan", False) and not a.ms_dis
        netD = self.model.discriminator1 if (a.ms_dis or ragan) else self.model.discriminator2
        adv, cls = self._generator_adv(netD, img_random, c_org, netD_real=self.model.discriminator2)
        if self.reparam:
            with ops.frozen(self.model.style_encoder):      # Es is not stepped in this phase (235-239)
                _, mu2, _ = self._encode_style(img_random, c_org)
            mu2_a, mu2_b = torch.split(mu2, B, dim=0)
            lat = [(self.l1_loss(mu2_a, z_sr), 10.0), (self.l1_loss(mu2_b, z_sr), 10.0)]
        else:
            with ops.frozen(self.model.style_encoder):
                z_rec, _, _ = self._encode_style(img_random, c_org)
            z_rec_a, _ = torch.split(z_rec, B, dim=0)
            lat = [(self.l1_loss(z_rec_a, z_sr), 20.0)]                 # a + a (base_model.py:419-420)
        loss_g, v, _ = ops.loss_sum([("l1_recon_z", lat, 1.0, 1.0), ("gan2", adv, 1.0, 1.0), ("gan2_cls", cls, 1.0, 1.0)])
        loss_g.backward()
        self._set_loss(**v)

    # ---- visuals / dispatch ---------------------------------------------------------------------------
    def compute_visuals(self):
        B = self.args.batch_size
        with torch.no_grad():
            img_fake, img_random, img_self = self.forward(self.img, self.c_org)
        fa, fb = torch.split(img_fake, B, dim=0)
        ra, rb = torch.split(img_random, B, dim=0)
        sa, sb = torch.split(img_self, B, dim=0)
        f32 = ops.to_nchw_f32
        ia, ib = torch.split(f32(self.img), B, dim=0)
        row1 = torch.cat((ia[0:1], f32(fb)[0:1], f32(rb)[0:1], f32(sa)[0:1]), dim=3)
        row2 = torch.cat((ib[0:1], f32(fa)[0:1], f32(ra)[0:1], f32(sb)[0:1]), dim=3)
        return torch.cat((row1, row2), dim=2)

    def normalize_image(self, x):
        return x[:, 0:3, :, :]

    def optimize_parameters(self, global_iter):
        if self._graph_mode():
            return self._optimize_graphed(global_iter)
        return self._optimize_eager(global_iter)

    def _optimize_eager(self, global_iter):
        if self.reducer.enabled:
            for opt in self.optimizer.values():
                opt.reset_pending()
        if self.args.use_dis_content and global_iter % self.args.d_iter != 0:
            self.update_content_discriminator(self.img, self.c_org)
            return
        self.update_discriminator(self.img, self.c_org)
        try:
            self.update_generator(self.img, self.c_org)
        finally:
            self._finish_deferred()             # (no-op unless update_generator raised before phase 4)
