"""FusedAdam: ``torch.optim.Adam`` semantics (L2-coupled weight decay, bias correction; reference
adain_model.py:57-61,68-71) executed as ONE HIP launch per optimizer over flat fp32 buffers.

Parameters, gradients and both moments of a network are views into four flat device buffers, so
 * ``step()`` is a single ``mt_adam_multi`` launch (no per-tensor kernels, no pointer chasing),
 * ``zero_grad()`` is one memset,
 * the data-parallel gradient exchange is one RCCL all-reduce of ``flat_grad`` per backward phase.
``state_dict()`` / ``load_state_dict()`` keep torch.optim.Adam's format so ``opt_{it}.ckpt`` files
interchange with the reference (model.py:70-100).
"""
import torch

from . import hip_ops as ops


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=(float(betas[0]), float(betas[1])), eps=eps, weight_decay=weight_decay)
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdam keeps ONE flat buffer with one set of hyper-parameters: pass a single parameter "
                             "group (the reference builds one Adam per network, adain_model.py:57-71)")
        self._flat = None
        self._step_count_mt = 0
        self.generation = 0          # bumped whenever the flat buffers are (re)built: captured graphs hold their addresses
        self._grads_clean = False    # True right after step(): the Adam kernel zeroed the gradients it consumed
        self._dev = None             # 16-byte device record {float lr; int32 step; float bc1; float bc2_sqrt}
        self._dev_lr = None

    # ---- flat storage ----
    def params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def _flatten(self):
        ps = self.params()
        dev = ps[0].device
        if dev.type != "cuda":
            raise RuntimeError("FusedAdam steps on the HIP device only (no CPU fallback)")
        offs, total = [], 0
        for p in ps:
            offs.append(total)
            total += (p.numel() + 3) & ~3          # keep every view 16-byte aligned
        fp = torch.zeros(total, dtype=torch.float32, device=dev)
        fg = torch.zeros(total, dtype=torch.float32, device=dev)
        fm = torch.zeros(total, dtype=torch.float32, device=dev)
        fv = torch.zeros(total, dtype=torch.float32, device=dev)
        step = 0
        for p, o in zip(ps, offs):
            n = p.numel()
            fp[o:o + n].copy_(p.data.reshape(-1))
            if p.grad is not None:
                fg[o:o + n].copy_(p.grad.reshape(-1))
            st = self.state.get(p, {})
            if "exp_avg" in st:
                fm[o:o + n].copy_(st["exp_avg"].reshape(-1))
                fv[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
                step = max(step, int(torch.as_tensor(st["step"]).item()))
            p.data = fp[o:o + n].view(p.shape)
            p.grad = fg[o:o + n].view(p.shape)
            self.state[p] = {"step": torch.tensor(float(step)), "exp_avg": fm[o:o + n].view(p.shape),
                             "exp_avg_sq": fv[o:o + n].view(p.shape)}
        self._step_count_mt = step
        self._offsets = offs
        self._flat = (fp, fg, fm, fv, total)
        self._ptrs = torch.tensor([fp.data_ptr(), fg.data_ptr(), fm.data_ptr(), fv.data_ptr()], dtype=torch.int64).to(dev)
        self._sizes = torch.tensor([total], dtype=torch.int64).to(dev)
        rec = torch.zeros(4, dtype=torch.float32)
        rec.view(torch.int32)[1] = step
        self._dev = rec.to(dev)
        self._dev_lr = None
        self._grads_clean = False
        self.generation += 1
        ops.bump_epoch(ps)

    def flat_grad(self):
        if self._flat is None:
            self._flatten()
        return self._flat[1]

    def flat_param(self):
        if self._flat is None:
            self._flatten()
        return self._flat[0]

    # ---- torch.optim API ----
    def zero_grad(self, set_to_none=False):
        # step() clears the gradient buffer in the pass that reads it: the memset is only needed when something wrote
        # gradients since (a backward pass whose optimizer step never came) or before the first step
        if not self._grads_clean:
            self.flat_grad().zero_()
        self._grads_clean = False       # (a backward pass is about to write)

    def reset_pending(self):
        """forget weight uses of graphs that were never run backward (see hip_ops._Conv.forward); called at the top of
        a training iteration, NOT by zero_grad: the encoder pass shared between the discriminator update and phase 3 is
        recorded before phase 3 zeroes the gradients"""
        for p in self.params():
            p._mt_pending = 0

    def grad_buckets(self, nbuckets=2, min_elems=None):
        """Split the flat gradient buffer at parameter boundaries into up to ``nbuckets`` contiguous ranges, LAST
        parameters first: a backward pass produces gradients in reverse parameter order, so bucket 0 (the tail of the
        buffer -- for the discriminators their 9.4 M-parameter last convolution) is complete almost as soon as the
        backward pass starts.  -> list of (start, end, [weight parameters whose readiness completes the bucket])."""
        if self._flat is None:
            self._flatten()
        if min_elems is None:           # (a collective below ~4 MB is latency bound: do not split small networks)
            import os
            min_elems = int(os.environ.get("MT_BUCKET_MIN_ELEMS", str(1 << 20)))
        ps, offs = self.params(), self._offsets
        total = self._flat[4]
        bounds = [total]
        want = max(total // max(int(nbuckets), 1), int(min_elems))
        acc = 0
        for i in range(len(ps) - 1, 0, -1):
            acc += offs[i + 1] - offs[i] if i + 1 < len(ps) else total - offs[i]
            # cut in front of a weight (dim > 1) so a bias stays with its weight
            if acc >= want and ps[i].dim() > 1 and len(bounds) < nbuckets:
                bounds.append(offs[i])
                acc = 0
        bounds.append(0)
        out = []
        for hi, lo in zip(bounds[:-1], bounds[1:]):
            if hi > lo:
                members = [p for p, o in zip(ps, offs) if lo <= o < hi]
                out.append((lo, hi, members))
        return out

    def sync_lr(self):
        """Write the current learning rate into the device record if the scheduler changed it (one tiny fill; a no-op
        on every other step).  A captured step graph calls this before each replay."""
        if self._flat is None:
            self._flatten()
        lr = float(self.param_groups[0]["lr"])
        if self._dev_lr != lr:
            self._dev[0:1].fill_(lr)
            self._dev_lr = lr

    def advance_host_step(self, n):
        """A replayed graph stepped this optimizer ``n`` times on the device: keep the host mirror (state_dict) in sync."""
        self._step_count_mt += int(n)

    @torch.no_grad()
    def step(self, closure=None):
        if self._flat is None:
            self._flatten()
        g = self.param_groups[0]
        self.sync_lr()
        self._step_count_mt += 1
        import ctypes as C
        from . import _lib as L
        b1, b2 = g["betas"]
        # learning rate, step count and bias corrections live in the device record (ticked by the call itself): no
        # launch argument changes from step to step, so the update can be replayed from a captured hipGraph
        # (4 reads -- parameter, gradient, both moments -- and 4 writes: the gradient is cleared in the same pass)
        with ops._hbm("adam_multi", int(self._flat[4]) * 4 * 8):
            L.check(L.load().mt_adam_multi_dev(C.c_void_p(self._ptrs.data_ptr()), C.c_void_p(self._sizes.data_ptr()), 1,
                                               self._flat[4], float(b1), float(b2), float(g["eps"]),
                                               float(g["weight_decay"]), C.c_void_p(self._dev.data_ptr()), 1, ops._stream()),
                    "mt_adam_multi_dev")
        self._grads_clean = True
        ops.bump_epoch(self.params())
        ops.repack_params(self.params())    # all cached weight images of this network, one launch

    def add_param_group(self, param_group):
        if getattr(self, "param_groups", None):
            raise ValueError("FusedAdam supports a single parameter group")
        super().add_param_group(param_group)

    def state_dict(self):
        if self._flat is not None:      # (before the first flatten the per-parameter 'step' entries are authoritative)
            for p in self.params():
                if p in self.state and "step" in self.state[p]:
                    self.state[p]["step"] = torch.tensor(float(self._step_count_mt))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [int(torch.as_tensor(s["step"]).item()) for s in self.state.values() if "step" in s]
        self._step_count_mt = max(steps) if steps else 0
        self._flat = None   # re-flatten (copies the loaded moments) at the next use
