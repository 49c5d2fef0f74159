"""Autograd-visible ops of the hot path, each a thin wrapper over one or more C-ABI entry
points of ``libmt_hip.so`` (``include/mt_api.h``).  PyTorch supplies device memory, the
stream and the autograd tape; all arithmetic happens in the HIP library.

Tensor convention ("canonical"): a 4-D activation is a logical NCHW tensor whose memory is
NHWC with the channel count padded to a multiple of 8 (strides ``(H*W*Cp, 1, W*Cp, Cp)``),
pad channels zero.  So module code reads exactly like the reference's (``torch.cat(..., 0)``,
``torch.split``, ``.size(1)``) while kernels see 16-byte aligned channel vectors.  Tensors in
any other layout/dtype are converted once on entry (``canon``).  2-D tensors (style codes,
class vectors, MLP activations) are plain fp32 row-major.

There is deliberately no CPU path: ops raise if handed a non-HIP tensor.
"""
import ctypes as C
import os
import contextlib

import torch

from . import _lib as L

_STATE = {"dtype": torch.float32}


def set_compute_dtype(dtype):
    """Storage dtype of activations: torch.float32 (parity path) or torch.bfloat16."""
    if dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(f"unsupported compute dtype {dtype}")
    _STATE["dtype"] = dtype


def compute_dtype():
    return _STATE["dtype"]


def padc(c):
    return (c + 7) & ~7


def _mt(dtype):
    return L.MT_BF16 if dtype == torch.bfloat16 else L.MT_F32


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """Raw handle of the current HIP stream (the C ABI takes the stream explicitly).  The private torch call is
    ~10x cheaper than ``torch.cuda.current_stream()`` (3.5k calls per training step)."""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_hip(t):
    if not t.is_cuda:
        raise RuntimeError("masterthesis_amd ops run on the HIP device only (there is no CPU fallback); "
                           f"got a tensor on {t.device}")


class _ZeroArena:
    """fp32 scratch that is zero when handed out: ONE memset per training-step scope instead of one fill kernel
    per statistics buffer (~250 of them per AdaINModel step).  Only transient buffers that are consumed inside
    the scope may come from here; outside a scope (inference, unit tests) ``take`` is a plain ``torch.zeros``."""

    def __init__(self):
        self.buf = None
        self.off = self.dirty = self.need = self.asked = self.depth = 0
        self.high = 0       # high-water mark of handed-out floats: the memset always covers it, so a step captured into a
                            # hipGraph re-zeroes everything ANY kind of step may have dirtied before it
        self.generation = 0  # bumped whenever the buffer is reallocated: a captured graph holds the OLD address

    def begin(self, device):
        self.depth += 1
        if self.depth > 1:
            return
        want = max(self.need, 1 << 18)
        if self.buf is None or self.buf.device != device or self.buf.numel() < want:
            self.buf = torch.zeros(int(want * 1.25), dtype=torch.float32, device=device)
            self.high = 0
            self.generation += 1
        elif self.dirty:
            self.buf[:self.dirty].zero_()
        self.off = self.dirty = self.asked = 0

    def end(self):
        self.depth -= 1
        if self.depth == 0:
            self.need = max(self.need, self.asked)
            self.high = max(self.high, self.off)
            self.dirty = self.high

    def take(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 63) & ~63
        self.asked += n_al
        if self.depth == 0 or self.buf is None or self.buf.device != device or self.off + n_al > self.buf.numel():
            return torch.zeros(shape, dtype=torch.float32, device=device)
        t = self.buf[self.off:self.off + n].view(shape)
        self.off += n_al
        return t


_arena = _ZeroArena()


def step_scope(fn):
    """Decorator for the model's ``update_*`` methods: zero-scratch hand-outs inside share one memset."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if dev is None:
            return fn(self, *a, **kw)
        _arena.begin(dev)
        try:
            return fn(self, *a, **kw)
        finally:
            _arena.end()
    return wrapped


def _zero_stats(shape, device):
    return _arena.take(shape, device)


def new_act(N, Cc, H, W, dtype, device, zero=False):
    """Allocate a canonical activation and return its logical NCHW view."""
    alloc = torch.zeros if zero else torch.empty
    buf = alloc((N, H, W, padc(Cc)), dtype=dtype, device=device)
    return buf.permute(0, 3, 1, 2)[:, :Cc]


def is_canonical(t):
    if t.dim() != 4 or not t.is_cuda:
        return False
    N, Cc, H, W = t.shape
    Cp = padc(Cc)
    want = (H * W * Cp, 1, W * Cp, Cp)
    for size, st, w in zip(t.shape, t.stride(), want):
        if size > 1 and st != w:
            return False
    return t.data_ptr() % 16 == 0


def canon(t, dtype=None):
    """Return ``t`` in canonical layout and compute dtype (no copy if it already is)."""
    dtype = dtype or compute_dtype()
    _need_hip(t)
    if t.dim() != 4:
        raise ValueError(f"expected a 4-D NCHW tensor, got shape {tuple(t.shape)}")
    if t.dtype == dtype and is_canonical(t):
        return t
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.float()
    N, Cc, H, W = t.shape
    out = new_act(N, Cc, H, W, dtype, t.device)
    sn, sc, sh, sw = t.stride()
    L.check(L.load().mt_to_nhwc(_mt(t.dtype), _ptr(t), sn, sc, sh, sw, _mt(dtype), _ptr(out), N, Cc, H, W,
                                _stream()), "mt_to_nhwc")
    return out


def to_nchw_f32(t):
    """Canonical activation -> dense fp32 NCHW tensor (API boundary / visualisation)."""
    t = canon(t.detach())
    N, Cc, H, W = t.shape
    out = torch.empty((N, Cc, H, W), dtype=torch.float32, device=t.device)
    L.check(L.load().mt_to_nchw_f32(_mt(t.dtype), _ptr(t), _ptr(out), N, Cc, H, W, _stream()), "mt_to_nchw_f32")
    return out


def _f32c(t):
    """fp32 contiguous device tensor (2-D / parameter operands)."""
    _need_hip(t)
    if t.dtype != torch.float32 or not t.is_contiguous():
        t = t.float().contiguous()
    return t


def _act_code(act):
    return {None: L.ACT_NONE, "none": L.ACT_NONE, "relu": L.ACT_RELU, "lrelu": L.ACT_LRELU,
            "tanh": L.ACT_TANH}[act]


# --------------------------------------------------------------------------------------
# packed-weight cache.  Parameters keep the reference layout (checkpoint compatible); the
# MFMA tile images are rebuilt only when a parameter changed (optimizer step / load).
# --------------------------------------------------------------------------------------
_EPOCH = {}


def bump_epoch(params):
    """Called by the optimizer after it rewrote parameter memory through raw pointers."""
    for p in params:
        k = p.data_ptr()
        _EPOCH[k] = _EPOCH.get(k, 0) + 1


def _pack_tag(owner, weight):
    return (owner._version, weight.data_ptr(), _EPOCH.get(weight.data_ptr(), 0), tuple(weight.shape))


def _get_pack(owner, weight, desc, which):
    """``owner`` is the nn.Parameter object: the cache lives ON it, so it dies with the parameter and can
    never be confused with another tensor that later reuses the same device address.  A stale image is
    re-packed IN PLACE (stable addresses: ``repack_params`` batches all of a network's images in one launch)."""
    cache = getattr(owner, "_mt_packs", None)
    if cache is None:
        cache = {}
        try:
            owner._mt_packs = cache
        except AttributeError:
            pass
    key = (which, desc.dtype, desc.transposed, desc.stride, desc.kh, desc.kw, desc.pad, desc.pad_mode)
    tag = _pack_tag(owner, weight)
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1]
    lib = L.load()
    nbytes = max(int(lib.mt_conv_pack_bytes(C.byref(desc), which)), 16)
    if hit is not None and hit[1].numel() == nbytes and hit[1].device == weight.device:
        pack = hit[1]
    else:
        pack = torch.empty((nbytes,), dtype=torch.uint8, device=weight.device)
    L.check(lib.mt_conv_pack(C.byref(desc), which, _ptr(weight), _ptr(pack), _stream()), "mt_conv_pack")
    cache[key] = (tag, pack, L.ConvDesc.from_buffer_copy(desc), which)
    return pack


_PACK_TABLES = {}
_PACK_TABLES_GEN = [0]      # bumped when the tables are dropped: a captured mt_conv_pack_multi_run reads the OLD table


def graph_epoch():
    """Changes whenever a device buffer that captured step graphs have baked in by ADDRESS was freed or reallocated (the
    zero arena, the batched-pack tables): part of the graph key, so a stale graph is never replayed."""
    return (_arena.generation, _PACK_TABLES_GEN[0])


def repack_params(params):
    """Re-pack every cached weight image of ``params`` in ONE launch (called by FusedAdam after its step: ~100
    single-image launches of ~6 us each otherwise).  The device-side table is built once per set of addresses."""
    items = []
    for p in params:
        cache = getattr(p, "_mt_packs", None)
        if not cache or not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            continue
        for key, (tag, pack, desc, which) in cache.items():
            items.append((p, key, pack, desc, which))
    if not items:
        return
    lib = L.load()
    sig = tuple((p.data_ptr(), pack.data_ptr(), which, bytes(desc)) for p, _, pack, desc, which in items)
    tab = _PACK_TABLES.get(sig)
    if tab is None:
        n = len(items)
        descs = (L.ConvDesc * n)(*[it[3] for it in items])
        whichs = (C.c_int * n)(*[it[4] for it in items])
        ws = (C.c_void_p * n)(*[it[0].data_ptr() for it in items])
        packs = (C.c_void_p * n)(*[it[2].data_ptr() for it in items])
        host = C.create_string_buffer(int(lib.mt_conv_pack_multi_table_bytes(n)))
        ne, nb = C.c_int(), C.c_int()
        L.check(lib.mt_conv_pack_multi_build(n, descs, whichs, ws, packs, host, C.byref(ne), C.byref(nb)),
                "mt_conv_pack_multi_build")
        dev = torch.frombuffer(host, dtype=torch.uint8).clone().to(items[0][0].device)
        if len(_PACK_TABLES) > 64:
            _PACK_TABLES.clear()
            _PACK_TABLES_GEN[0] += 1
        tab = (dev, ne.value, nb.value)
        _PACK_TABLES[sig] = tab
    L.check(lib.mt_conv_pack_multi_run(_ptr(tab[0]), tab[1], tab[2], _stream()), "mt_conv_pack_multi_run")
    for p, key, pack, desc, which in items:
        p._mt_packs[key] = (_pack_tag(p, p), pack, desc, which)


# --------------------------------------------------------------------------------------
# optional HIP-event timing of selected forward-conv launches (used by bench.py's roofline leg)
# --------------------------------------------------------------------------------------
_KTIMER = {"match": None, "events": []}
_FUSE_WGRAD_ACC = [False]
_DETERMINISTIC = [os.environ.get("MT_DETERMINISTIC", "0") == "1"]


def set_deterministic(on):
    """Bit-reproducible runs: every reduction whose summation order depends on scheduling (the fp32 atomics of the
    fused convolution-statistics epilogue, of the loss reductions and of the thin 1x1 weight gradient) is replaced by
    its fixed-order variant.  Costs one extra statistics pass per normalised convolution.  Also MT_DETERMINISTIC=1."""
    _DETERMINISTIC[0] = bool(on)


def deterministic():
    return _DETERMINISTIC[0]


def set_fused_grad_accumulation(on):
    """When on, conv weight/bias gradients are ADDED straight into ``param.grad`` by the wgrad epilogue
    (no temporary + torch add); autograd then receives None for them.  Requires ``param.grad`` to exist
    (FusedAdam's flat gradient views) -- only valid for ``loss.backward()`` style training."""
    _FUSE_WGRAD_ACC[0] = bool(on)


def _fused_grad_target(p):
    """param.grad if gradients may be accumulated into it in place by the backward kernel, else None"""
    g = None if (p is None or not p.is_leaf) else p.grad
    ok = (_FUSE_WGRAD_ACC[0] and g is not None and g.dtype == torch.float32 and g.is_contiguous()
          and g.shape == p.shape and g.is_cuda)
    return g if ok else None


def kernel_timer_start(match):
    """Time every conv forward whose descriptor satisfies ``match(desc)`` with HIP events recorded on the
    launch stream."""
    _KTIMER["match"] = match
    _KTIMER["events"] = []


def kernel_timer_stop():
    """-> list of (milliseconds, batch N of the launch) per timed launch (synchronises)."""
    _KTIMER["match"] = None
    torch.cuda.synchronize()
    out = [(a.elapsed_time(b), n) for a, b, n in _KTIMER["events"]]
    _KTIMER["events"] = []
    return out


_OPLOG = {"on": False, "events": []}


def oplog_start():
    """Record HIP events around every conv forward / data-gradient / weight-gradient call (tools/layer_table.py)."""
    _OPLOG["on"] = True
    _OPLOG["events"] = []


def oplog_stop():
    """-> list of (kind, descriptor tuple, milliseconds)."""
    _OPLOG["on"] = False
    torch.cuda.synchronize()
    out = [(k, d, a.elapsed_time(b)) for k, d, a, b in _OPLOG["events"]]
    _OPLOG["events"] = []
    return out


_HBMLOG = {"on": False, "events": []}


def hbm_timer_start():
    """Record HIP events around the HBM-bound kernels of the step (normalisation statistics / apply passes, Adam) together
    with their algorithmic bytes (bench.py's ``hbm_kernels`` block; SURVEY 8d: GB/s of those kernels reported separately)."""
    _HBMLOG["on"] = True
    _HBMLOG["events"] = []


def hbm_timer_stop():
    """-> {kernel: (calls, total ms, total algorithmic bytes)} (synchronises)."""
    _HBMLOG["on"] = False
    torch.cuda.synchronize()
    out = {}
    for name, nbytes, a, b in _HBMLOG["events"]:
        e = out.setdefault(name, [0, 0.0, 0])
        e[0] += 1
        e[1] += a.elapsed_time(b)
        e[2] += nbytes
    _HBMLOG["events"] = []
    return {k: tuple(v) for k, v in out.items()}


class _hbm:
    """with _hbm(name, algorithmic bytes): <one launch>"""
    def __init__(self, name, nbytes):
        self.on = _HBMLOG["on"]
        self.name, self.nbytes = name, nbytes

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.on:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _HBMLOG["events"].append((self.name, int(self.nbytes), self.e0, e1))


class _oplog:
    def __init__(self, kind, desc, extra=()):
        self.on = _OPLOG["on"]
        if self.on:
            self.key = (kind, tuple(getattr(desc, f[0]) for f in desc._fields_) + tuple(extra))

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.on:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _OPLOG["events"].append((self.key[0], self.key[1], self.e0, e1))


# --------------------------------------------------------------------------------------
# convolution family
# --------------------------------------------------------------------------------------
class GradLink:
    """Carries the skip-connection gradient of a residual block from the normalisation that adds the residual (its backward
    runs first) to the block's first convolution (its backward runs last): ``y = norm(conv2(..conv1(x)..), res=x)``.  With a
    link, the norm's backward parks ``dy`` here instead of returning it as the gradient of ``res``, and conv1's backward
    hands it to ``mt_conv_bwd_data_add`` -- the sum ``dgrad + dy`` is formed in the data-gradient GEMM's epilogue (or by the
    library's add where that kernel does not apply) instead of by autograd's own accumulation pass over the three tensors
    (``aten::add``: 0.6 ms per step).  Both ends must see the SAME tensor x (so both or neither need its gradient)."""
    __slots__ = ("g",)

    def __init__(self):
        self.g = None


class StatsLink:
    """Hands the statistics pass of a normalisation BACKWARD to the kernel that produces its gradient.  A norm whose output
    y is consumed by a residual block only -- the block's first convolution plus the skip -- gets its whole gradient
    ``dy = dgrad(conv1) + skip`` out of conv1's data-gradient epilogue (GradLink), and a norm in front of a plain
    convolution gets ``dy = dgrad(conv)``; the sums the norm's backward needs, {sum g, sum g * x} with g = dy * act'(..),
    can be taken right there (``mt_conv_bwd_data_ex``) instead of in a pass of their own over dy and x (``mt_nc_stats_bwd``:
    63 launches, 1.7 ms per step).  The norm's forward fills ``x, scale, shift, act, slope`` and hangs the link on its
    output tensor (``y._mt_stats_link``); a consumer that is the ONLY path of y's gradient passes it to ``conv2d(...,
    bwd_stats=link)``; the convolution's backward leaves ``sums`` here when the kernel that ran could produce them, and the
    norm's backward then skips its statistics pass.  Any other situation falls back to the norm's own pass."""
    __slots__ = ("x", "scale", "shift", "act", "slope", "sums")

    def __init__(self):
        self.x = self.scale = self.shift = self.sums = None
        self.act, self.slope = 0, 0.0


# Off by default: measured same-box (round 3), the data-gradient epilogue -- the one moment when all 256 CUs hit HBM
# together -- grows by 17 us per K1 launch for the extra tile read, the stand-alone statistics pass it replaces costs 27 us
# in isolation, and the step does not move (36.63-36.82 vs 36.65-36.71 ms): the pass it removes was also what pulled x into
# the Infinity Cache for the apply pass that follows.  MT_STATS_LINK=1 / set_stats_link(True) turn it on.
_STATS_LINK_ON = [os.environ.get("MT_STATS_LINK", "0") == "1"]


def set_stats_link(on):
    _STATS_LINK_ON[0] = bool(on)


# One-pass backward of InstanceNorm / AdaIN (mt_norm_bwd_onepass: statistics + coefficients + apply in one launch, dy and x read
# once; planes of 1024 / 2048 / 4096 pixels in bf16).  MT_NORM_ONEPASS=0 / set_norm_onepass(False): the three-launch backward.
_NORM_ONEPASS_ON = [os.environ.get("MT_NORM_ONEPASS", "1") != "0"]


def set_norm_onepass(on):
    _NORM_ONEPASS_ON[0] = bool(on)


_ONEPASS_SYNC = {}
_ONEPASS_LAST = {}          # device index -> torch.cuda.Stream of the last one-pass launch
_ONEPASS_RESERVE = [0]      # compute units left to a kernel that overlaps the backward pass (a live gradient exchange)
_ONEPASS_SPIN = [int(os.environ.get("MT_OP_SPIN", "0"))]     # poll bound of the wait (0: the library's 2^19)
_ONEPASS_IMAGES = 8192      # images per launch the counters are sized for (a bigger batch takes the three launches)
_DEV_STATUS = {}


def set_onepass_reserve(cus):
    """Compute units to leave to OTHER kernels that may be resident during a backward pass (the RCCL kernels of a gradient
    exchange that overlaps it): the one-pass norm backward then only takes problems whose slices per image fit the rest."""
    _ONEPASS_RESERVE[0] = max(int(cus), 0)


def device_status(dev):
    """int32 [4] status words of ``dev``, zero while all is well; kernels that can fail on the device (the bounded wait of the
    one-pass norm backward) set a bit instead of failing silently.  ``models.Model.sync_losses`` copies them to the host together
    with the loss scalars and calls ``raise_on_device_status`` -- no extra synchronisation.  Never allocated during a graph capture
    (the zero-fill would be a graph node and the words would sit in the graph's private pool)."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _DEV_STATUS.get(idx)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        st = torch.zeros(4, dtype=torch.int32, device=dev)
        _DEV_STATUS[idx] = st
    return st


def raise_on_device_status(words, dev=None):
    """``words``: host copy of ``device_status`` (a sequence of ints).  Raises RuntimeError if a kernel reported a failure and
    clears the device words so that the next step starts clean."""
    if not words or not int(words[0]):
        return
    if dev is not None:
        st = _DEV_STATUS.get(dev.index if dev.index is not None else torch.cuda.current_device())
        if st is not None:
            st.zero_()
    what = []
    if int(words[0]) & 1:
        what.append(f"one-pass norm backward: workgroup {int(words[1]) - 1} gave up waiting for the other slices of its image "
                    "(another kernel that waits for sibling workgroups shares the device, or the launch did not fit it); "
                    "the gradients of this step are NaN.  MT_NORM_ONEPASS=0 selects the three-launch backward")
    raise RuntimeError("masterthesis_amd device status: " + "; ".join(what or [f"status words {list(words)}"]))


def check_device_status(dev=None):
    """Synchronising form (tests, tools): read the status words now and raise if one is set."""
    dev = dev or torch.device("cuda", torch.cuda.current_device())
    st = device_status(dev)
    if st is not None:
        raise_on_device_status(st.cpu().tolist(), dev)


def _onepass_sync(dev, n):
    """Per-(device, stream) arrive / leave counters of mt_norm_bwd_onepass ([2][N] uint32, zero; the kernel leaves them zero), sized
    once for ``_ONEPASS_IMAGES`` images so that a captured graph keeps using the same address and nothing is ever (re)allocated
    while a capture is in progress.  None: no buffer can be had right now -> the caller takes the three launches."""
    if n > _ONEPASS_IMAGES:
        return None
    cur = torch.cuda.current_stream(dev)
    key = (dev.index, cur.cuda_stream)
    buf = _ONEPASS_SYNC.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        buf = torch.zeros(2 * _ONEPASS_IMAGES, dtype=torch.int32, device=dev)
        _ONEPASS_SYNC[key] = buf
    # single-stream contract of the kernel (include/mt_api.h): two launches must never be resident together.  A launch from
    # another stream than the previous one is ordered behind everything that stream holds (no per-launch event on the usual
    # one-stream path; at the start of a capture the device has been synchronised, and a capture cannot wait on the outside)
    last = _ONEPASS_LAST.get(dev.index)
    if last is not None and last.cuda_stream != cur.cuda_stream and not torch.cuda.is_current_stream_capturing():
        cur.wait_stream(last)
    _ONEPASS_LAST[dev.index] = cur
    return buf


def _onepass_slices(lib, mt, mode, N, HW, Cp, act):
    """slices per image if the one-pass backward takes this problem (0: three launches)"""
    cap = int(lib.mt_norm_bwd_onepass_capacity()) - _ONEPASS_RESERVE[0]
    if cap < 1:
        return 0
    slices = C.c_int(0)
    if not lib.mt_norm_bwd_onepass_ok(mt, mode, N, HW, Cp, act, cap, C.byref(slices)):
        return 0
    return slices.value


def stats_link_of(t):
    """the StatsLink a normalisation hung on its output tensor, if any (and fused statistics are allowed at all)"""
    return None if (_DETERMINISTIC[0] or not _STATS_LINK_ON[0]) else getattr(t, "_mt_stats_link", None)


_DESC_INFO = {}
_DESC_CACHE_ON = os.environ.get("MT_DESC_CACHE", "1") != "0"      # (A/B runs)


def _desc_info(lib, desc):
    """Pure functions of a convolution descriptor (output size, workspace sizes, whether the statistics epilogue applies), asked
    of the library once per descriptor: five ctypes calls per convolution call otherwise -- the small layers of the
    discriminators are host-bound in the eager step."""
    key = (bytes(desc), lib.mt_kernel_variant_epoch())       # (workspace sizes follow the kernel-variant switches)
    info = _DESC_INFO.get(key) if _DESC_CACHE_ON else None
    if info is None:
        ho, wo = C.c_int(), C.c_int()
        L.check(lib.mt_conv_out_hw(C.byref(desc), C.byref(ho), C.byref(wo)), "mt_conv_out_hw")
        info = (ho.value, wo.value, bool(lib.mt_conv_fwd_stats_fused(C.byref(desc))), int(lib.mt_conv_fwd_ws_bytes(C.byref(desc))),
                int(lib.mt_conv_bwd_data_ws_bytes(C.byref(desc))), int(lib.mt_conv_bwd_weight_ws_bytes(C.byref(desc))))
        if len(_DESC_INFO) > 4096:
            _DESC_INFO.clear()
        _DESC_INFO[key] = info
    return info


class _Conv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, cfg):
        stride, pad, pad_mode, act, slope, transposed, out_pad, want_stats, bias_grad = cfg[:9]
        lib = L.load()
        dt = compute_dtype()
        x = canon(x)
        owner = weight
        weight = _f32c(weight.detach())
        N, Ci, H, W = x.shape
        if transposed:
            Ciw, Co = weight.shape[0], weight.shape[1]
        else:
            Co, Ciw = weight.shape[0], weight.shape[1]
        if Ciw != Ci:
            raise RuntimeError(f"conv: input has {Ci} channels, weight expects {Ciw}")
        desc = L.ConvDesc(_mt(dt), int(transposed), N, H, W, Ci, Co, weight.shape[2], weight.shape[3],
                          stride, pad, pad_mode, out_pad, act, slope)
        ho, wo, stats_ok, nws_fwd, _, _ = _desc_info(lib, desc)
        y = new_act(N, Co, ho, wo, dt, x.device)
        pack = _get_pack(owner, weight, desc, L.PACK_FWD)
        b = None if bias is None else _f32c(bias.detach())
        # normalisation statistics of the output are accumulated (atomics) in the GEMM epilogue where the shape allows
        # it; otherwise (and in deterministic mode) the norm layer runs its own reproducible statistics pass
        want_stats = bool(want_stats and not _DETERMINISTIC[0] and stats_ok)
        sums = _zero_stats((N, padc(Co), 2), x.device) if want_stats else None
        timed = _KTIMER["match"] is not None and _KTIMER["match"](desc)
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        with _oplog("fwd", desc, (int(want_stats),)):
            if want_stats:
                L.check(lib.mt_conv_fwd_stats(C.byref(desc), _ptr(x), _ptr(pack), _ptr(b), _ptr(y), _ptr(sums),
                                              _stream()), "mt_conv_fwd_stats")
            else:
                nws = nws_fwd                                           # > 0 only for split-K shapes
                ws = torch.empty((nws,), dtype=torch.uint8, device=x.device) if nws else None
                L.check(lib.mt_conv_fwd_ex(C.byref(desc), _ptr(x), _ptr(pack), _ptr(b), _ptr(y), _ptr(ws), nws,
                                           _stream()), "mt_conv_fwd_ex")
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _KTIMER["events"].append((e0, e1, N))
        ctx.desc = desc
        ctx.owner = owner
        ctx.link = cfg[10] if len(cfg) > 10 else None
        ctx.bstats = cfg[11] if len(cfg) > 11 else None
        ctx.bias_owner = bias
        # uses of this weight whose weight gradient is still to come in the backward pass under construction: when the
        # count returns to zero the parameter's gradient is final (it is accumulated straight into param.grad), which
        # is what lets the data-parallel exchange of a bucket start INSIDE the backward pass (set_grad_ready_hook)
        # (grad mode is always off INSIDE Function.forward: the caller's mode travels in cfg[9])
        ctx.counted = bool(len(cfg) > 9 and cfg[9] and owner.requires_grad and owner.is_leaf)
        if ctx.counted:
            owner._mt_pending = getattr(owner, "_mt_pending", 0) + 1
        # the statistics output never carries a gradient: without this autograd materialises a zero tensor for it
        # on every backward call (57 fill kernels per step)
        ctx.set_materialize_grads(False)
        ctx.has_bias = bias is not None and bias_grad
        ctx.save_for_backward(x, weight, y if act != L.ACT_NONE else None)
        if cfg[7]:                         # the caller asked for statistics: always a pair (sums may be None)
            if sums is not None:
                ctx.mark_non_differentiable(sums)
            return y, sums
        return y

    @staticmethod
    def backward(ctx, dy, dsums=None):
        lib = L.load()
        x, weight, y = ctx.saved_tensors
        desc = ctx.desc
        if dy is None:
            _grad_use_done(ctx)
            return None, None, None, None
        dy = canon(dy)
        dx = dw = db = None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        gb = _fused_grad_target(ctx.bias_owner) if want_b else None
        bias_done = False
        if desc.act != L.ACT_NONE:
            dz = new_act(*dy.shape, dy.dtype, dy.device)
            Cp = padc(dy.shape[1])
            npix = dy.shape[0] * dy.shape[2] * dy.shape[3]
            if want_b:
                # activation derivative and bias gradient in ONE pass over dy
                nws = int(lib.mt_act_bwd_bias_ws_bytes(Cp))
                ws = torch.empty((nws,), dtype=torch.uint8, device=dy.device)
                if gb is None:
                    db = torch.empty((desc.Co,), dtype=torch.float32, device=dy.device)
                L.check(lib.mt_act_bwd_bias(desc.dtype, _ptr(dy), _ptr(y), _ptr(dz), npix, Cp, desc.Co, desc.act, desc.slope,
                                            _ptr(gb if gb is not None else db), int(gb is not None), _ptr(ws), nws, _stream()),
                        "mt_act_bwd_bias")
                bias_done = True
            else:
                L.check(lib.mt_act_bwd(desc.dtype, _ptr(dy), _ptr(y), _ptr(dz), npix * Cp, desc.act, desc.slope, _stream()),
                        "mt_act_bwd")
            dy = dz
        if ctx.needs_input_grad[0]:
            pack = _get_pack(ctx.owner, weight, desc, L.PACK_BWD_DATA)
            nws = _desc_info(lib, desc)[4]
            ws = torch.empty((max(nws, 16),), dtype=torch.uint8, device=dy.device)
            dx = new_act(*x.shape, dy.dtype, dy.device)
            skip = None
            if ctx.link is not None and ctx.link.g is not None:
                skip, ctx.link.g = ctx.link.g, None
                if skip.shape != dx.shape or skip.dtype != dx.dtype:
                    raise RuntimeError("residual gradient link: the parked gradient does not match the block input")
            bl = ctx.bstats
            if bl is not None and (bl.x is None or _DETERMINISTIC[0] or bl.x.shape != dx.shape or bl.x.dtype != dx.dtype):
                bl = None
            with _oplog("dgrad", desc):
                if bl is not None:
                    sums2 = _zero_stats((x.shape[0], padc(x.shape[1]), 2), dy.device)
                    bs = L.BwdStats(_ptr(bl.x), _ptr(bl.scale), _ptr(bl.shift), _ptr(sums2), int(bl.act), float(bl.slope))
                    done = C.c_int(0)
                    L.check(lib.mt_conv_bwd_data_ex(C.byref(desc), _ptr(dy), _ptr(pack), _ptr(dx), _ptr(skip), C.byref(bs),
                                                    C.byref(done), _ptr(ws), nws, _stream()), "mt_conv_bwd_data_ex")
                    bl.sums = sums2 if done.value else None
                elif skip is not None:
                    L.check(lib.mt_conv_bwd_data_add(C.byref(desc), _ptr(dy), _ptr(pack), _ptr(dx), _ptr(skip), _ptr(ws), nws,
                                                     _stream()), "mt_conv_bwd_data_add")
                else:
                    L.check(lib.mt_conv_bwd_data(C.byref(desc), _ptr(dy), _ptr(pack), _ptr(dx), _ptr(ws), nws, _stream()),
                            "mt_conv_bwd_data")
        need_b = want_b and not bias_done           # bias gradient still to be taken by the weight-gradient call
        if ctx.needs_input_grad[1] or need_b:

            gw = _fused_grad_target(ctx.owner) if ctx.needs_input_grad[1] else None
            gbw = gb if need_b else None
            if need_b and gw is not None and gbw is not None and _wgrad_rows_parkable(ctx, desc):
                # a layer of the row walker's shared launches whose bias gradient rides on the weight-gradient call (no fused
                # activation in front of it, e.g. ConvTranspose2d -> LayerNorm): the bias gradient now, by itself (the column
                # sums of dy: the same launch mt_conv_bwd_weight would make first), the weight gradient with the others
                nws = _desc_info(lib, desc)[5]
                ws = torch.empty((nws,), dtype=torch.uint8, device=dy.device)
                ns = C.c_int(0)
                L.check(lib.mt_conv_bwd_weight_partial(C.byref(desc), _ptr(x), _ptr(dy), _ptr(gbw), _ptr(ws), nws, 1, 0, C.byref(ns),
                                                       _stream()), "mt_conv_bwd_weight_partial")
                need_b, gbw = False, None
            if gw is not None and not need_b and (_wgrad_rows_defer(ctx, desc, x, dy, gw)
                                                  or _wgrad_group_defer(ctx, desc, x, dy, gw)
                                                  or _wgrad_share_defer(ctx, desc, x, dy, gw)
                                                  or _wgrad_finish_defer(ctx, desc, x, dy, gw)):
                return dx, None, db, None          # launched with its group (or at the end of this backward pass)
            nws = _desc_info(lib, desc)[5]
            ws = torch.empty((nws,), dtype=torch.uint8, device=dy.device)
            if gw is not None and (gbw is not None or not need_b):
                # accumulate in place; autograd gets None for both
                with _oplog("wgrad", desc, (int(need_b),)):
                    L.check(lib.mt_conv_bwd_weight(C.byref(desc), _ptr(x), _ptr(dy), _ptr(gw), _ptr(gbw), _ptr(ws), nws,
                                                   1, _stream()), "mt_conv_bwd_weight")
            else:
                dw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
                if need_b:
                    db = torch.empty((desc.Co,), dtype=torch.float32, device=dy.device)
                with _oplog("wgrad", desc, (int(need_b),)):
                    L.check(lib.mt_conv_bwd_weight(C.byref(desc), _ptr(x), _ptr(dy), _ptr(dw), _ptr(db if need_b else None),
                                                   _ptr(ws), nws, 0, _stream()), "mt_conv_bwd_weight")
        _grad_use_done(ctx)
        return dx, dw, db, None


# ---- grouped weight gradients ------------------------------------------------------------------------------------------
# Weight gradients are leaves of the backward pass: nothing in it waits for them.  The 256x256 weight-gradient kernel splits the
# pixel reduction 28 ways to fill the chip for ONE problem of the dominant layer (66 MB of fp32 slabs written and read back); up
# to four problems of the same descriptor share one launch instead (mt_conv_bwd_weight_group: 28 / G splits each).  A problem
# waits in a per-descriptor queue until its group is full; what is left when the backward pass ends goes out from an autograd
# engine callback (queued with the first deferred problem), so param.grad is complete when backward() returns.
# MT_WGRAD_GROUP=0 / set_wgrad_group(False): every weight gradient is launched where autograd reaches it.
_WGRAD_GROUP_ON = [os.environ.get("MT_WGRAD_GROUP", "1") != "0"]
_WGRAD_GROUP_CAP = int(os.environ.get("MT_WGRAD_GROUP_MAX", "8"))       # (A/B runs: cap the group size)
_WGRAD_QUEUE = {"pending": {}, "armed": None, "gmax": {}}      # armed: id of the autograd graph task whose callback will flush


def set_wgrad_group(on):
    flush_wgrad_groups()
    _WGRAD_GROUP_ON[0] = bool(on)


def _desc_key(desc):
    return bytes(desc)


def _wgrad_group_defer(ctx, desc, x, dy, gw):
    if not _WGRAD_GROUP_ON[0]:
        return False
    key = _desc_key(desc)
    ckey = (key, L.load().mt_kernel_variant_epoch())          # (the answer follows the kernel-variant switches, as _desc_info)
    gmax = _WGRAD_QUEUE["gmax"].get(ckey)
    if gmax is None:
        gmax = min(int(L.load().mt_conv_bwd_weight_group_max(C.byref(desc))), _WGRAD_GROUP_CAP)
        _WGRAD_QUEUE["gmax"][ckey] = gmax
    if gmax < 2:
        return False
    if not _wgrad_arm():
        return False                                 # not inside an engine run (backward called by hand): nothing would flush
    q = _WGRAD_QUEUE["pending"].setdefault(key, [])
    q.append((ctx, desc, x, dy, gw, _park_event()))
    if len(q) >= gmax:
        _wgrad_group_launch(_WGRAD_QUEUE["pending"].pop(key))
    return True


def _park_event():
    """(stream, event) of the moment a weight-gradient problem was parked: its operands are ready on THAT stream"""
    st = torch.cuda.current_stream()
    return (st.cuda_stream, st.record_event())


def _wait_parked(items):
    """the launching stream waits for the operands of problems parked from other streams, and the caching allocator learns that
    they are in use on the launching stream too (it would otherwise hand x / dy out again as soon as the parking stream is
    done with them, while the grouped launch still reads them: ADVICE r3)"""
    cur = torch.cuda.current_stream()
    for it in items:
        if it[5][0] != cur.cuda_stream:
            cur.wait_event(it[5][1])
            it[2].record_stream(cur)
            it[3].record_stream(cur)


def _wgrad_group_launch(items):
    _wait_parked(items)
    lib = L.load()
    desc = items[0][1]
    G = len(items)

    dev = items[0][2].device
    if G == 1:
        ctx, desc, x, dy, gw = items[0][:5]
        nws = int(lib.mt_conv_bwd_weight_ws_bytes(C.byref(desc)))
        ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
        with _oplog("wgrad", desc, (0,)):
            L.check(lib.mt_conv_bwd_weight(C.byref(desc), _ptr(x), _ptr(dy), _ptr(gw), None, _ptr(ws), nws, 1, _stream()),
                    "mt_conv_bwd_weight")
    else:
        nws = int(lib.mt_conv_bwd_weight_group_ws_bytes(C.byref(desc), G))
        if nws == 0:                         # this group size does not fit one launch (e.g. three of four): split it
            _wgrad_group_launch(items[:G // 2])
            _wgrad_group_launch(items[G // 2:])
            return
        ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
        xs = (C.c_void_p * G)(*[it[2].data_ptr() for it in items])
        dys = (C.c_void_p * G)(*[it[3].data_ptr() for it in items])
        gws = (C.c_void_p * G)(*[it[4].data_ptr() for it in items])
        with _oplog("wgrad", desc, (0, G)):
            L.check(lib.mt_conv_bwd_weight_group(C.byref(desc), G, xs, dys, gws, _ptr(ws), nws, 1, _stream()),
                    "mt_conv_bwd_weight_group")
    for it in items:
        _grad_use_done(it[0])


def _wgrad_arm():
    """-> False outside an autograd engine run; else makes sure this run's end-of-backward callback will flush the queues"""
    get_task = getattr(torch._C, "_current_graph_task_id", None)
    if get_task is None:                             # (a torch without the hook: never park, launch where autograd reaches it)
        return False
    task = get_task()
    if task < 0:
        return False
    if _WGRAD_QUEUE["armed"] != task:
        if _WGRAD_QUEUE["pending"]:                  # left over from a backward pass that died before its callback ran
            drop_wgrad_groups()
        _WGRAD_QUEUE["armed"] = task
        torch.autograd.Variable._execution_engine.queue_callback(flush_wgrad_groups)
    return True


def _wgrad_share_defer(ctx, desc, x, dy, gw):
    """Several uses of ONE weight in this backward pass (the three scales of a MultiScaleDiscriminator; an encoder applied twice):
    every use still runs its own split GEMM, but the slabs go behind each other into one workspace and ONE slab sum adds them
    into param.grad -- the 1024 -> 2048 layer of the multi-scale discriminators reads and re-writes its 134 MB gradient once
    instead of three times.  A use waits until the weight's last use of the pass has arrived (ctx.counted bookkeeping)."""
    if not _WGRAD_GROUP_ON[0] or not ctx.counted:
        return False
    owner = ctx.owner
    key = ("owner", id(owner))
    parked = _WGRAD_QUEUE["pending"].get(key)
    if parked is None and getattr(owner, "_mt_pending", 0) <= 1:
        return False                                 # the only use (left) of this weight
    skey = ("slab", _desc_key(desc), L.load().mt_kernel_variant_epoch())
    sb = _WGRAD_QUEUE["gmax"].get(skey)
    if sb is None:
        sb = int(L.load().mt_conv_bwd_weight_slab_bytes(C.byref(desc)))
        _WGRAD_QUEUE["gmax"][skey] = sb
    if sb == 0 or not _wgrad_arm():
        return False
    if parked is None:
        parked = _WGRAD_QUEUE["pending"].setdefault(key, [])
    parked.append((ctx, desc, x, dy, gw, _park_event()))
    if len(parked) >= getattr(owner, "_mt_pending", 0):
        _wgrad_shared_launch(_WGRAD_QUEUE["pending"].pop(key))
    return True


# Batched slab sums (round 4).  The split GEMM of a weight gradient runs where autograd reaches it, its slab sum -- 91 launches of
# 3-16 us per step, mostly a launch's fixed cost -- waits for the end of the backward pass, where ONE mt_conv_bwd_weight_finish_multi
# per (up to 64) weights adds them into param.grad.  Only without a live gradient-ready hook (a data-parallel bucket wants its
# gradients as early as possible).  MT_WGRAD_FINISH_BATCH=0 / set_wgrad_finish_batch(False): every slab sum behind its GEMM.
_WGRAD_FINISH_ON = [os.environ.get("MT_WGRAD_FINISH_BATCH", "1") != "0"]


def set_wgrad_finish_batch(on):
    flush_wgrad_groups()
    _WGRAD_FINISH_ON[0] = bool(on)


def _wgrad_finish_defer(ctx, desc, x, dy, gw):
    if not _WGRAD_FINISH_ON[0] or not _WGRAD_GROUP_ON[0] or getattr(ctx.owner, "_mt_ready_hook", None) is not None:
        return False
    lib = L.load()
    skey = ("slab", _desc_key(desc), lib.mt_kernel_variant_epoch())
    sb = _WGRAD_QUEUE["gmax"].get(skey)
    if sb is None:
        sb = int(lib.mt_conv_bwd_weight_slab_bytes(C.byref(desc)))
        _WGRAD_QUEUE["gmax"][skey] = sb
    if sb == 0 or not _wgrad_arm():
        return False
    nws = _desc_info(lib, desc)[5]
    ws = torch.empty((nws,), dtype=torch.uint8, device=dy.device)
    ns = C.c_int(0)
    with _oplog("wgrad", desc, (0,)):
        L.check(lib.mt_conv_bwd_weight_partial(C.byref(desc), _ptr(x), _ptr(dy), None, _ptr(ws), nws, 1, 1, C.byref(ns), _stream()),
                "mt_conv_bwd_weight_partial")
    _WGRAD_QUEUE["pending"].setdefault(("finish",), []).append((ctx, desc, ws, ns.value, gw, _park_event()))
    return True


# Shared launches of the row walker (round 4; csrc/wgrad_rows_kernel.hip).  The 3x3 layers of the encoders and the decoder at 64-256
# channels each fill the chip only by cutting their pixel reduction ~256 ways (38 MB of slabs per layer); parked until the end of
# the backward pass, all of them go out in ONE launch per stride class (mt_conv_bwd_weight_rows_multi) with a share of the compute
# units each, and one batched slab sum.  Not with a live gradient-ready hook (a data-parallel bucket wants its gradients early).
# MT_WGRAD_ROWS_MULTI=0 / set_wgrad_rows_multi(False): every such layer launches where autograd reaches it.
_WGRAD_ROWS_ON = [os.environ.get("MT_WGRAD_ROWS_MULTI", "1") != "0"]
_WGRAD_ROWS_MAX = 48


def set_wgrad_rows_multi(on):
    flush_wgrad_groups()
    _WGRAD_ROWS_ON[0] = bool(on)


def _wgrad_rows_parkable(ctx, desc):
    """would _wgrad_rows_defer park this weight gradient?"""
    if not _WGRAD_ROWS_ON[0] or not _WGRAD_GROUP_ON[0] or getattr(ctx.owner, "_mt_ready_hook", None) is not None:
        return False
    lib = L.load()
    rkey = ("rows_ok", _desc_key(desc), lib.mt_kernel_variant_epoch())
    ok = _WGRAD_QUEUE["gmax"].get(rkey)
    if ok is None:
        ok = bool(lib.mt_conv_bwd_weight_rows_ok(C.byref(desc)))
        _WGRAD_QUEUE["gmax"][rkey] = ok
    return bool(ok and _wgrad_arm())


def _wgrad_rows_defer(ctx, desc, x, dy, gw):
    if not _wgrad_rows_parkable(ctx, desc):
        return False
    q = _WGRAD_QUEUE["pending"].setdefault(("rows",), [])
    q.append((ctx, desc, x, dy, gw, _park_event()))
    if len(q) >= _WGRAD_ROWS_MAX:
        _wgrad_rows_launch(_WGRAD_QUEUE["pending"].pop(("rows",)))
    return True


def _wgrad_rows_launch(items):
    """every parked row-walker problem in one library call; uses of one weight next to each other (one slab sum per weight)"""
    _wait_parked(items)
    order = {}
    for it in items:
        order.setdefault(it[4].data_ptr(), len(order))
    items = sorted(items, key=lambda it: order[it[4].data_ptr()])          # (stable: first-use order of the weights)
    lib = L.load()
    n = len(items)
    descs = (L.ConvDesc * n)(*[it[1] for it in items])
    nws = int(lib.mt_conv_bwd_weight_rows_multi_ws_bytes(n, descs))
    if nws == 0:
        raise RuntimeError("mt_conv_bwd_weight_rows_multi_ws_bytes: " + (lib.mt_last_error() or b"").decode())
    ws = torch.empty((nws,), dtype=torch.uint8, device=items[0][2].device)
    xs = (C.c_void_p * n)(*[it[2].data_ptr() for it in items])
    dys = (C.c_void_p * n)(*[it[3].data_ptr() for it in items])
    gws = (C.c_void_p * n)(*[it[4].data_ptr() for it in items])
    macs = 0
    if _OPLOG["on"]:                                   # (tools/layer_table.py: the launch's work is the sum over its layers)
        for it in items:
            d = it[1]
            pix = d.N * d.H * d.W if d.transposed else d.N * ((d.H + 2 * d.pad - d.kh) // d.stride + 1) * ((d.W + 2 * d.pad - d.kw) // d.stride + 1)
            macs += pix * d.Ci * d.Co * d.kh * d.kw
    with _oplog("wgrad_rows", items[0][1], (0, n, macs)):
        L.check(lib.mt_conv_bwd_weight_rows_multi(n, descs, xs, dys, gws, _ptr(ws), nws, 1, _stream()),
                "mt_conv_bwd_weight_rows_multi")
    for it in items:
        _grad_use_done(it[0])


def _wgrad_finish_launch(items):
    """the slab sums of every parked (desc, slabs, param.grad) in as few launches as the library needs"""
    cur = torch.cuda.current_stream()
    for it in items:
        if it[5][0] != cur.cuda_stream:
            cur.wait_event(it[5][1])
            it[2].record_stream(cur)
    lib = L.load()
    n = len(items)
    descs = (L.ConvDesc * n)(*[it[1] for it in items])
    wss = (C.c_void_p * n)(*[it[2].data_ptr() for it in items])
    nsl = (C.c_int * n)(*[it[3] for it in items])
    gws = (C.c_void_p * n)(*[it[4].data_ptr() for it in items])
    with _oplog("wgrad_sum", items[0][1], (0, n)):
        L.check(lib.mt_conv_bwd_weight_finish_multi(n, descs, wss, nsl, gws, 1, _stream()), "mt_conv_bwd_weight_finish_multi")
    for it in items:
        _grad_use_done(it[0])


def _wgrad_shared_launch(items):
    """partial GEMMs of every use into one workspace, one slab sum (items: uses of one weight, any geometry)"""
    if len(items) == 1:
        return _wgrad_group_launch(items)
    _wait_parked(items)
    lib = L.load()
    dev = items[0][2].device
    gw = items[0][4]
    sb = int(lib.mt_conv_bwd_weight_slab_bytes(C.byref(items[0][1])))
    sizes = [int(lib.mt_conv_bwd_weight_ws_bytes(C.byref(it[1]))) for it in items]
    total = sum((n + sb - 1) // sb * sb for n in sizes)
    ws = torch.empty((total,), dtype=torch.uint8, device=dev)
    used = 0
    ns = C.c_int(0)
    for it in items:
        ctx, desc, x, dy = it[:4]
        off = used * sb
        with _oplog("wgrad", desc, (0,)):
            L.check(lib.mt_conv_bwd_weight_partial(C.byref(desc), _ptr(x), _ptr(dy), None, C.c_void_p(ws.data_ptr() + off),
                                                   total - off, 1, 1, C.byref(ns), _stream()), "mt_conv_bwd_weight_partial")
        used += ns.value
    with _oplog("wgrad_sum", items[0][1], (0,)):
        L.check(lib.mt_conv_bwd_weight_finish(C.byref(items[0][1]), _ptr(ws), used, _ptr(gw), 1, _stream()),
                "mt_conv_bwd_weight_finish")
    for it in items:
        _grad_use_done(it[0])


def flush_wgrad_groups():
    """launch every deferred weight gradient (autograd engine callback at the end of a backward pass; also safe to call by hand)"""
    _WGRAD_QUEUE["armed"] = None
    pend, _WGRAD_QUEUE["pending"] = _WGRAD_QUEUE["pending"], {}
    for key, items in pend.items():
        if key == ("finish",):                       # slab sums waiting for the end of the pass
            _wgrad_finish_launch(items)
        elif key == ("rows",):                       # 3x3 layers that share launches of the row walker
            _wgrad_rows_launch(items)
        elif isinstance(key, tuple):                 # ("owner", id): uses of one weight
            _wgrad_shared_launch(items)
        else:
            _wgrad_group_launch(items)


def drop_wgrad_groups():
    """Forget every parked weight gradient WITHOUT launching it: the backward pass that parked them died before its end-of-pass
    callback ran, so they belong to a step that failed -- adding them into param.grad during the next pass would mix two steps
    (ADVICE r3).  The use counters of their weights are released so that the gradient-ready bookkeeping starts clean."""
    _WGRAD_QUEUE["armed"] = None
    pend, _WGRAD_QUEUE["pending"] = _WGRAD_QUEUE["pending"], {}
    for items in pend.values():
        for it in items:
            ctx = it[0]
            if ctx.counted:
                ctx.owner._mt_pending = max(getattr(ctx.owner, "_mt_pending", 1) - 1, 0)
            link = getattr(ctx, "link", None)
            if link is not None:
                link.g = None                        # a skip gradient parked by the dead pass


def _grad_use_done(ctx):
    if not ctx.counted:
        return
    owner = ctx.owner
    owner._mt_pending = getattr(owner, "_mt_pending", 1) - 1
    if owner._mt_pending < 0:
        # more weight-gradient launches than recorded uses: a bucket would have been exchanged too early
        hooked = getattr(owner, "_mt_ready_hook", None) is not None
        owner._mt_pending = 0
        if hooked:
            raise RuntimeError("gradient-ready bookkeeping underflow: a weight was used in a graph recorded before its "
                               "use counter was reset (FusedAdam.reset_pending)")
    elif owner._mt_pending == 0:
        hook = getattr(owner, "_mt_ready_hook", None)
        if hook is not None:
            hook(owner)


def set_grad_ready_hook(param, fn):
    """``fn(param)`` is called from the backward pass right after the LAST weight-gradient launch of ``param`` in the
    current graph has been enqueued (None clears it).  Only meaningful with fused gradient accumulation."""
    param._mt_ready_hook = fn


def conv2d(x, weight, bias=None, stride=1, pad=0, pad_mode="zero", act=None, slope=0.01, stats=False, bias_grad=True,
           grad_link=None, bwd_stats=None):
    """act(conv2d(pad(x)) + bias): nn.ReflectionPad2d/zero pad + nn.Conv2d (+ activation).

    stats=True: also return the per-(image, channel) {sum, sum of squares} of the output, accumulated in the
    GEMM epilogue, for the normalisation layer that follows (pass it as ``sums=``).
    bias_grad=False: the bias is added but receives no gradient -- for a conv directly followed by an
    affine-free InstanceNorm the bias gradient is identically zero (the reference computes round-off there).
    grad_link: a ``GradLink`` shared with the normalisation that adds ``x`` back as a residual (see there).
    bwd_stats: the ``StatsLink`` of the normalisation that produced ``x``, when this convolution (+ grad_link) is the only
    path of x's gradient: its data gradient then also delivers that norm's backward statistics."""
    pm = L.PAD_REFLECT if (pad_mode == "reflect" and pad > 0) else L.PAD_ZERO
    return _Conv.apply(x, weight, bias, (stride, pad, pm, _act_code(act), float(slope), False, 0, bool(stats),
                                         bool(bias_grad), torch.is_grad_enabled(), grad_link, bwd_stats))


def conv_transpose2d(x, weight, bias=None, stride=1, pad=0, out_pad=0, act=None, slope=0.01):
    """nn.ConvTranspose2d (+ activation); weight layout [Cin, Cout, kh, kw]."""
    return _Conv.apply(x, weight, bias, (stride, pad, L.PAD_ZERO, _act_code(act), float(slope), True, out_pad, False,
                                         True, torch.is_grad_enabled()))


class _SpectralScale(torch.autograd.Function):
    """weight_orig / sigma with (u, v, sigma) as constants of the call: dW = (G - <G, W/sigma> u v^T) / sigma."""
    @staticmethod
    def forward(ctx, weight, u, v, sigma):
        w = _f32c(weight.detach())
        out = torch.empty_like(w)
        L.check(L.load().mt_sn_scale_fwd(_ptr(w), _ptr(sigma), _ptr(out), w.numel(), _stream()), "mt_sn_scale_fwd")
        ctx.save_for_backward(out, u, v, sigma)
        return out

    @staticmethod
    def backward(ctx, g):
        weff, u, v, sigma = ctx.saved_tensors
        lib = L.load()
        g = _f32c(g)
        rows, cols = weff.shape[0], weff.numel() // weff.shape[0]
        nws = int(lib.mt_sn_ws_bytes(rows, cols))
        ws = torch.empty((nws,), dtype=torch.uint8, device=g.device)
        dw = torch.empty_like(weff)
        L.check(lib.mt_sn_scale_bwd(_ptr(g), _ptr(weff), _ptr(u), _ptr(v), _ptr(sigma), _ptr(dw), rows, cols, _ptr(ws),
                                    nws, _stream()), "mt_sn_scale_bwd")
        return dw, None, None, None


def spectral_norm_weight(weight_orig, u, v, training=True, n_power_iterations=1, eps=1e-12):
    """torch.nn.utils.spectral_norm's ``compute_weight`` for a Conv2d weight (dim 0 = rows of the matrix view;
    reference functions.py:113-121): in training mode ``n_power_iterations`` power iterations update the buffers
    ``u`` [Cout] and ``v`` [Cin*kh*kw] IN PLACE without gradient; returns weight_orig / sigma (fp32 OIHW) whose
    gradient reaches weight_orig through both the scaling and sigma = u . W v."""
    _need_hip(weight_orig)
    lib = L.load()
    w = _f32c(weight_orig.detach())
    rows, cols = w.shape[0], w.numel() // w.shape[0]
    if u.numel() != rows or v.numel() != cols or u.dtype != torch.float32 or v.dtype != torch.float32:
        raise RuntimeError(f"spectral_norm_weight: u/v of {u.numel()}/{v.numel()} elements for a {rows}x{cols} weight")
    nws = int(lib.mt_sn_ws_bytes(rows, cols))
    ws = torch.empty((nws,), dtype=torch.uint8, device=w.device)
    sigma = torch.empty((1,), dtype=torch.float32, device=w.device)
    L.check(lib.mt_sn_power_iter(_ptr(w), _ptr(u), _ptr(v), _ptr(sigma), rows, cols,
                                 int(n_power_iterations) if training else 0, float(eps), _ptr(ws), nws, _stream()),
            "mt_sn_power_iter")
    # the iteration continues in place on the next call: the graph keeps this call's vectors (torch clones them too)
    if weight_orig.requires_grad and torch.is_grad_enabled():
        u, v = u.clone(), v.clone()
    return _SpectralScale.apply(weight_orig, u, v, sigma)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, grad_on=False):
        x = _f32c(x)
        w = _f32c(weight.detach())
        b = None if bias is None else _f32c(bias.detach())
        n, i = x.shape
        o = w.shape[0]
        if w.shape[1] != i:
            raise RuntimeError(f"linear: input has {i} features, weight expects {w.shape[1]}")
        y = torch.empty((n, o), dtype=torch.float32, device=x.device)
        L.check(L.load().mt_linear_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(y), n, i, o, L.ACT_NONE, _stream()),
                "mt_linear_fwd")
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.owners = (weight, bias)
        ctx.owner = weight
        ctx.counted = bool(grad_on and weight.requires_grad and weight.is_leaf)   # (see _Conv.forward)
        if ctx.counted:
            weight._mt_pending = getattr(weight, "_mt_pending", 0) + 1
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _f32c(dy)
        n, i = x.shape
        o = w.shape[0]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        want_b = need_w and ctx.has_bias
        gw = _fused_grad_target(ctx.owners[0]) if need_w else None
        gb = _fused_grad_target(ctx.owners[1]) if want_b else None
        if need_w and gw is not None and (gb is not None or not want_b):
            # accumulate straight into param.grad; autograd receives None for both
            L.check(L.load().mt_linear_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(gw), _ptr(gb), n, i, o, 1,
                                           _stream()), "mt_linear_bwd")
            _grad_use_done(ctx)
            return dx, None, None, None
        dw = torch.empty_like(w) if need_w else None
        db = torch.empty((o,), dtype=torch.float32, device=x.device) if want_b else None
        L.check(L.load().mt_linear_bwd(_ptr(x), _ptr(w), _ptr(dy), _ptr(dx), _ptr(dw), _ptr(db), n, i, o, 0,
                                       _stream()), "mt_linear_bwd")
        _grad_use_done(ctx)
        return dx, (dw if ctx.needs_input_grad[1] else None), db, None


def linear(x, weight, bias=None):
    return _Linear.apply(x, weight, bias, torch.is_grad_enabled())


class _LinearGroup(torch.autograd.Function):
    """G nn.Linear layers of equal shape on ONE input in one launch (forward) / two launches (backward: dx summed over
    the groups, dw / db per group) -- the four AdaIN projections of a decoder."""

    @staticmethod
    def forward(ctx, x, grad_on, *params):
        G = len(params) // 2
        ws, bs = params[:G], params[G:]
        x = _f32c(x)
        wd = [_f32c(w.detach()) for w in ws]
        bd = [None if b is None else _f32c(b.detach()) for b in bs]
        n, i = x.shape
        o = wd[0].shape[0]
        if any(tuple(w.shape) != (o, i) for w in wd):
            raise RuntimeError("linear_grouped: all layers must share the shape [out, in] and match the input")
        ys = [torch.empty((n, o), dtype=torch.float32, device=x.device) for _ in range(G)]
        arr = lambda ts: (C.c_void_p * G)(*[None if t is None else t.data_ptr() for t in ts])   # noqa: E731
        L.check(L.load().mt_linear_group_fwd(_ptr(x), arr(wd), arr(bd), arr(ys), G, n, i, o, _stream()),
                "mt_linear_group_fwd")
        ctx.save_for_backward(x, *wd)
        ctx.G, ctx.owners_w, ctx.owners_b = G, ws, bs
        ctx.counted_list = []
        for w in ws:
            cnt = bool(grad_on and w.requires_grad and w.is_leaf)
            ctx.counted_list.append(cnt)
            if cnt:
                w._mt_pending = getattr(w, "_mt_pending", 0) + 1
        return tuple(ys)

    @staticmethod
    def backward(ctx, *dys):
        x, *wd = ctx.saved_tensors
        G = ctx.G
        n, i = x.shape
        o = wd[0].shape[0]
        dys = [torch.zeros((n, o), dtype=torch.float32, device=x.device) if d is None else _f32c(d) for d in dys]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        need_w = [ctx.needs_input_grad[2 + g] for g in range(G)]
        need_b = [ctx.owners_b[g] is not None and ctx.needs_input_grad[2 + G + g] for g in range(G)]
        gw = [_fused_grad_target(ctx.owners_w[g]) if need_w[g] else None for g in range(G)]
        gb = [_fused_grad_target(ctx.owners_b[g]) if need_b[g] else None for g in range(G)]
        fused = all((not need_w[g] or gw[g] is not None) and (not need_b[g] or gb[g] is not None) for g in range(G))
        if fused:
            dw, db = gw, gb
        else:
            dw = [torch.empty_like(wd[g]) if need_w[g] else None for g in range(G)]
            db = [torch.empty((o,), dtype=torch.float32, device=x.device) if need_b[g] else None for g in range(G)]
        arr = lambda ts: (C.c_void_p * G)(*[None if t is None else t.data_ptr() for t in ts])   # noqa: E731
        L.check(L.load().mt_linear_group_bwd(_ptr(x), arr(wd), arr(dys), _ptr(dx), arr(dw), arr(db), G, n, i, o, int(fused),
                                             _stream()), "mt_linear_group_bwd")
        for g in range(G):
            if ctx.counted_list[g]:
                _grad_use_done(_UseToken(ctx.owners_w[g]))
        if fused:
            return (dx, None) + (None,) * (2 * G)
        return (dx, None) + tuple(dw) + tuple(db)


class _UseToken:
    """adapter for _grad_use_done: one counted use of ``owner``"""
    counted = True

    def __init__(self, owner):
        self.owner = owner


def linear_grouped(x, layers):
    """[nn.Linear-like (weight, bias), ...] of equal shape applied to the same input -> list of outputs, one launch."""
    out = []
    for k in range(0, len(layers), 8):              # (the launch carries up to 8 pointer sets)
        ws = [w for w, _ in layers[k:k + 8]]
        bs = [b for _, b in layers[k:k + 8]]
        out += list(_LinearGroup.apply(x, torch.is_grad_enabled(), *ws, *bs))
    return out


# --------------------------------------------------------------------------------------
# normalisation family
# --------------------------------------------------------------------------------------
class _Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gb, gamma, beta, res, sums, cfg):
        mode, act, slope, eps = cfg[:4]
        bn = cfg[4] if len(cfg) > 4 else None           # BatchNorm: (running_mean, running_var, momentum, training)
        lib = L.load()
        x = canon(x)
        N, Cc, H, W = x.shape
        Cp, HW = padc(Cc), H * W
        dev = x.device
        mt = _mt(x.dtype)
        nparts = 1
        if bn is not None and not bn[3]:
            sums = None                                  # eval mode: running statistics, no pass over x
        elif sums is None:
            nparts = int(lib.mt_nc_stats_parts(mt, N, HW, Cp))
            sums = torch.empty((N, nparts, Cp, 2), dtype=torch.float32, device=dev)     # every element is written
            L.check(lib.mt_nc_stats(mt, _ptr(x), _ptr(sums), N, HW, Cp, _stream()), "mt_nc_stats")
        elif tuple(sums.shape) != (N, Cp, 2):
            raise RuntimeError(f"norm: precomputed statistics have shape {tuple(sums.shape)}, expected {(N, Cp, 2)}")
        coef = torch.empty((4, N, Cp), dtype=torch.float32, device=dev)  # scale, shift, mean, rstd
        gbc = None if gb is None else _f32c(gb.detach())
        gm = None if gamma is None else _f32c(gamma.detach())
        bt = None if beta is None else _f32c(beta.detach())
        if gbc is not None and tuple(gbc.shape) != (N, 2 * Cc):
            raise RuntimeError(f"adain: expected style projection of shape {(N, 2 * Cc)}, got {tuple(gbc.shape)}")
        r = None if res is None else canon(res)
        y = new_act(N, Cc, H, W, x.dtype, dev)
        if mode != L.NORM_BATCH and sums is not None and nparts == 1 and Cp <= 2048:
            # complete per-image statistics (the convolution's epilogue produced them): finalize + apply in ONE launch
            with _hbm("norm_apply_fused", N * HW * Cp * x.element_size() * (2 if r is None else 3)):
                L.check(lib.mt_norm_apply_fused(mt, mode, _ptr(x), _ptr(sums), _ptr(gbc), _ptr(gm), _ptr(bt), _ptr(r), _ptr(y),
                                                _ptr(coef), N, HW, Cc, Cp, act, slope, eps, _stream()), "mt_norm_apply_fused")
        else:
            if mode == L.NORM_BATCH:
                rm, rv, momentum, training = bn
                L.check(lib.mt_bn_finalize(_ptr(sums), _ptr(gm), _ptr(bt), _ptr(rm), _ptr(rv), float(momentum), eps,
                                           int(training), _ptr(coef[0]), _ptr(coef[1]), _ptr(coef[2]), _ptr(coef[3]), N, HW,
                                           Cc, Cp, nparts, _stream()), "mt_bn_finalize")
            else:
                L.check(lib.mt_norm_finalize(mode, _ptr(sums), _ptr(gbc), _ptr(gm), _ptr(bt), _ptr(coef[0]), _ptr(coef[1]),
                                             _ptr(coef[2]), _ptr(coef[3]), N, HW, Cc, Cp, eps, nparts, _stream()),
                        "mt_norm_finalize")
            L.check(lib.mt_scale_shift_act(mt, _ptr(x), _ptr(coef[0]), _ptr(coef[1]), _ptr(r), _ptr(y), N, HW, Cp, act,
                                           slope, _stream()), "mt_scale_shift_act")
        ctx.cfg = cfg
        ctx.save_for_backward(x, coef, gbc, gm)
        ctx.shapes = (None if gamma is None else gamma.shape, None if beta is None else beta.shape)
        sl = cfg[6] if len(cfg) > 6 else None          # StatsLink created by the wrapper (it hangs it on the output tensor)
        if sl is not None:
            sl.x, sl.scale, sl.shift, sl.act, sl.slope = x, coef[0], coef[1], act, slope
        ctx.slink = sl
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        mode, act, slope, eps = ctx.cfg[:4]
        x, coef, gbc, gm = ctx.saved_tensors
        dy = canon(dy)
        N, Cc, H, W = x.shape
        Cp, HW = padc(Cc), H * W
        dev = x.device
        mt = _mt(x.dtype)
        sl = ctx.slink
        linked = sl is not None and sl.sums is not None and tuple(sl.sums.shape) == (N, Cp, 2)
        nsl = 0
        if _NORM_ONEPASS_ON[0] and not linked and ctx.needs_input_grad[0]:
            nsl = _onepass_slices(lib, mt, mode, N, HW, Cp, act)
        sync = _onepass_sync(dev, N) if nsl else None
        status = device_status(dev) if sync is not None else None
        if status is not None:
            if sl is not None:
                sl.sums = None
            dgb = torch.empty_like(gbc) if mode == L.NORM_ADAIN else None
            dgamma = dbeta = None
            if mode == L.NORM_LAYER and gm is not None:
                dgamma = torch.empty((Cc,), dtype=torch.float32, device=dev)
                dbeta = torch.empty((Cc,), dtype=torch.float32, device=dev)
                dgb = torch.empty((N, 2, Cc), dtype=torch.float32, device=dev)          # per-image terms of dgamma / dbeta
            dx = new_act(N, Cc, H, W, x.dtype, dev)
            part = torch.empty((N, nsl, Cp, 2), dtype=torch.float32, device=dev)
            with _hbm("norm_bwd_onepass", N * HW * Cp * x.element_size() * 3):
                L.check(lib.mt_norm_bwd_onepass(mt, mode, _ptr(dy), _ptr(x), _ptr(coef[0]), _ptr(coef[1]), _ptr(coef[2]),
                                                _ptr(coef[3]), _ptr(gbc), _ptr(dgb), _ptr(gm), _ptr(dgamma), _ptr(dbeta),
                                                _ptr(dx), _ptr(part), _ptr(sync), _ptr(status), _ONEPASS_SPIN[0], N, HW, Cc,
                                                Cp, act, slope, _stream()), "mt_norm_bwd_onepass")
            if mode == L.NORM_LAYER:
                gshape, bshape = ctx.shapes
                if dgamma is not None:
                    dgamma, dbeta = dgamma.view(gshape), dbeta.view(bshape)
                return dx, None, dgamma, dbeta, None, None, None
            dres = dy if ctx.needs_input_grad[4] else None
            link = ctx.cfg[5] if len(ctx.cfg) > 5 else None
            if link is not None and dres is not None:
                link.g, dres = dres, None
            return dx, dgb, None, None, dres, None, None
        if linked:
            # the kernel that produced dy took the sums in its epilogue (StatsLink): one partial row per image
            sums2, nparts = sl.sums, 1
            sl.sums = None
        else:
            nparts = int(lib.mt_nc_stats_parts(mt, N, HW, Cp))
            sums2 = torch.empty((N, nparts, Cp, 2), dtype=torch.float32, device=dev)
            with _hbm("nc_stats_bwd", N * HW * Cp * x.element_size() * 2):
                L.check(lib.mt_nc_stats_bwd(mt, _ptr(dy), _ptr(x), _ptr(coef[0]), _ptr(coef[1]), _ptr(sums2), N, HW, Cp, act,
                                            slope, _stream()), "mt_nc_stats_bwd")
        cc = torch.empty((3, N, Cp), dtype=torch.float32, device=dev)
        dgb = torch.empty_like(gbc) if mode == L.NORM_ADAIN else None
        dgamma = dbeta = None
        if mode in (L.NORM_LAYER, L.NORM_BATCH) and gm is not None:
            dgamma = torch.empty((Cc,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((Cc,), dtype=torch.float32, device=dev)
            if mode == L.NORM_LAYER:
                dgb = torch.empty((N, 2, Cc), dtype=torch.float32, device=dev)      # per-image terms of dgamma / dbeta
        if mode == L.NORM_BATCH:
            L.check(lib.mt_bn_bwd_finalize(_ptr(sums2), _ptr(coef[2]), _ptr(coef[3]), _ptr(gm), _ptr(cc[0]), _ptr(cc[1]),
                                           _ptr(cc[2]), _ptr(dgamma), _ptr(dbeta), int(ctx.cfg[4][3]), N, HW, Cc, Cp,
                                           nparts, _stream()), "mt_bn_bwd_finalize")
        else:
            L.check(lib.mt_norm_bwd_finalize(mode, _ptr(sums2), _ptr(coef[2]), _ptr(coef[3]), _ptr(gbc), _ptr(gm),
                                             _ptr(cc[0]), _ptr(cc[1]), _ptr(cc[2]), _ptr(dgb), _ptr(dgamma),
                                             _ptr(dbeta), N, HW, Cc, Cp, nparts, _stream()), "mt_norm_bwd_finalize")
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new_act(N, Cc, H, W, x.dtype, dev)
            with _hbm("norm_bwd_apply", N * HW * Cp * x.element_size() * 3):
                L.check(lib.mt_norm_bwd_apply(mt, _ptr(dy), _ptr(x), _ptr(coef[0]), _ptr(coef[1]), _ptr(cc[0]),
                                              _ptr(cc[1]), _ptr(cc[2]), _ptr(dx), N, HW, Cp, act, slope, _stream()),
                        "mt_norm_bwd_apply")
        gshape, bshape = ctx.shapes
        if dgamma is not None:
            dgamma = dgamma.view(gshape)
            dbeta = dbeta.view(bshape)
        dres = dy if ctx.needs_input_grad[4] else None
        link = ctx.cfg[5] if len(ctx.cfg) > 5 else None
        if link is not None and dres is not None:
            link.g, dres = dres, None       # the block's first convolution adds it inside its data-gradient epilogue
        return dx, (dgb if mode == L.NORM_ADAIN else None), dgamma, dbeta, dres, None, None


def instance_norm_act(x, act=None, slope=0.01, res=None, eps=1e-5, sums=None, res_link=None):
    """act(InstanceNorm2d(affine=False)(x)) (+ res); ``sums`` = statistics from conv2d(..., stats=True)"""
    sl = StatsLink() if (_STATS_LINK_ON[0] and torch.is_grad_enabled() and not _DETERMINISTIC[0]) else None
    y = _Norm.apply(x, None, None, None, res, sums, (L.NORM_INSTANCE, _act_code(act), float(slope), float(eps), None,
                                                     res_link if res is not None else None, sl))
    if sl is not None:
        y._mt_stats_link = sl
    return y


def adain_act(x, gb, act=None, slope=0.01, res=None, eps=1e-5, sums=None, res_link=None):
    """act((1 + gb[:, :C]) * IN(x) + gb[:, C:]) (+ res)  -- reference norm.py:29-33"""
    sl = StatsLink() if (_STATS_LINK_ON[0] and torch.is_grad_enabled() and not _DETERMINISTIC[0]) else None
    y = _Norm.apply(x, gb, None, None, res, sums, (L.NORM_ADAIN, _act_code(act), float(slope), float(eps), None,
                                                   res_link if res is not None else None, sl))
    if sl is not None:
        y._mt_stats_link = sl
    return y


def batch_norm_act(x, gamma, beta, running_mean, running_var, training=True, momentum=0.1, act=None, slope=0.01,
                   eps=1e-5, sums=None):
    """act(nn.BatchNorm2d(affine, running statistics)(x)): batch statistics + in-place momentum update of the running
    buffers in training mode, the running buffers in eval mode (reference functions.py:14-15)."""
    return _Norm.apply(x, None, gamma, beta, None, sums,
                       (L.NORM_BATCH, _act_code(act), float(slope), float(eps),
                        (running_mean, running_var, float(momentum), bool(training))))


def layer_norm_act(x, gamma, beta, act=None, slope=0.01, eps=1e-5):
    """reference LayerNorm (per-sample over C,H,W; per-channel affine) + activation"""
    return _Norm.apply(x, None, gamma, beta, None, None, (L.NORM_LAYER, _act_code(act), float(slope), float(eps)))


# --------------------------------------------------------------------------------------
# elementwise / pooling / layout
# --------------------------------------------------------------------------------------
def _numel_padded(t):
    return t.shape[0] * t.shape[2] * t.shape[3] * padc(t.shape[1])


class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        if x.dim() == 4:
            x = canon(x)
            y = new_act(*x.shape, x.dtype, x.device)
            n, mt = _numel_padded(x), _mt(x.dtype)
        else:
            x = _f32c(x)
            y = torch.empty_like(x)
            n, mt = x.numel(), L.MT_F32
        L.check(L.load().mt_act_fwd(mt, _ptr(x), _ptr(y), n, act, slope, _stream()), "mt_act_fwd")
        ctx.cfg = (act, slope, n, mt)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        act, slope, n, mt = ctx.cfg
        if y.dim() == 4:
            dy = canon(dy)
            dx = new_act(*y.shape, y.dtype, y.device)
        else:
            dy = _f32c(dy)
            dx = torch.empty_like(y)
        L.check(L.load().mt_act_bwd(mt, _ptr(dy), _ptr(y), _ptr(dx), n, act, slope, _stream()), "mt_act_bwd")
        return dx, None, None


def activation(x, act, slope=0.01):
    return _Act.apply(x, _act_code(act), float(slope))


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = canon(a), canon(b)
        if a.shape != b.shape:
            raise RuntimeError(f"add: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        y = new_act(*a.shape, a.dtype, a.device)
        L.check(L.load().mt_add(_mt(a.dtype), _ptr(a), _ptr(b), _ptr(y), _numel_padded(a), _stream()), "mt_add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


class _NoiseAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, seed, offset):
        x = canon(x)
        y = new_act(*x.shape, x.dtype, x.device)
        L.check(L.load().mt_gaussian_noise_add(_mt(x.dtype), _ptr(x), _ptr(y), _numel_padded(x), seed, offset,
                                               _stream()), "mt_gaussian_noise_add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None


class _MulScale(torch.autograd.Function):
    """x * mask * scale with a constant mask (dropout): the same product is the gradient."""
    @staticmethod
    def forward(ctx, x, mask, scale):
        x = canon(x)
        y = new_act(*x.shape, x.dtype, x.device)
        L.check(L.load().mt_mul_scale(_mt(x.dtype), _ptr(x), _ptr(mask), _ptr(y), _numel_padded(x), scale, _stream()),
                "mt_mul_scale")
        ctx.save_for_backward(mask)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy = canon(dy)
        dx = new_act(*dy.shape, dy.dtype, dy.device)
        L.check(L.load().mt_mul_scale(_mt(dy.dtype), _ptr(dy), _ptr(mask), _ptr(dx), _numel_padded(dy), ctx.scale,
                                      _stream()), "mt_mul_scale")
        return dx, None, None


def bernoulli_mask(shape, keep, seed, offset, device):
    """0/1 mask ~ Bernoulli(keep) as a canonical activation (Philox counter kernel)."""
    N, Cc, H, W = shape
    m = new_act(N, Cc, H, W, compute_dtype(), device)
    L.check(L.load().mt_bernoulli_mask(_mt(m.dtype), _ptr(m), N * H * W, Cc, padc(Cc), float(keep), seed, offset,
                                       _stream()), "mt_bernoulli_mask")
    return m


def dropout(x, mask, p=0.5):
    """nn.Dropout(p) in training mode with the given keep-mask (canonical 0/1 activation): x * mask / (1 - p)."""
    return _MulScale.apply(x, canon(mask), 1.0 / (1.0 - p))


def gaussian_noise_add(x, seed, offset):
    """x + N(0,1) drawn on device (Philox4x32-10, counter = element index + offset)."""
    if padc(x.shape[1]) != x.shape[1]:
        raise RuntimeError("gaussian_noise_add needs a channel count that is a multiple of 8")
    return _NoiseAdd.apply(x, int(seed), int(offset))


class _NoiseAddDev(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, state):
        x = canon(x)
        y = new_act(*x.shape, x.dtype, x.device)
        lib = L.load()
        L.check(lib.mt_gaussian_noise_add_dev(_mt(x.dtype), _ptr(x), _ptr(y), _numel_padded(x), _ptr(state), _stream()),
                "mt_gaussian_noise_add_dev")
        L.check(lib.mt_rng_advance(_ptr(state), _stream()), "mt_rng_advance")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None


def gaussian_noise_add_dev(x, state):
    """x + N(0,1) from the device-resident generator ``state`` (int64 [2] = {seed, draw counter}); the counter advances
    in stream order, so the call is hipGraph-capturable and every replay draws fresh noise."""
    if padc(x.shape[1]) != x.shape[1]:
        raise RuntimeError("gaussian_noise_add needs a channel count that is a multiple of 8")
    return _NoiseAddDev.apply(x, state)


def bernoulli_mask_dev(shape, keep, state, device):
    """0/1 mask ~ Bernoulli(keep) from the device-resident generator state (see gaussian_noise_add_dev)."""
    N, Cc, H, W = shape
    m = new_act(N, Cc, H, W, compute_dtype(), device)
    lib = L.load()
    L.check(lib.mt_bernoulli_mask_dev(_mt(m.dtype), _ptr(m), N * H * W, Cc, padc(Cc), float(keep), _ptr(state), _stream()),
            "mt_bernoulli_mask_dev")
    L.check(lib.mt_rng_advance(_ptr(state), _stream()), "mt_rng_advance")
    return m


class _Pool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        lib = L.load()
        x = canon(x)
        N, Cc, H, W = x.shape
        if kind == 2:
            Ho, Wo, fn = H // 2, W // 2, lib.mt_avgpool2_fwd
        else:
            Ho, Wo, fn = (H - 1) // 2 + 1, (W - 1) // 2 + 1, lib.mt_avgpool3s2_fwd
        y = new_act(N, Cc, Ho, Wo, x.dtype, x.device)
        L.check(fn(_mt(x.dtype), _ptr(x), _ptr(y), N, H, W, padc(Cc), _stream()), "mt_avgpool_fwd")
        ctx.cfg = (kind, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        kind, (N, Cc, H, W) = ctx.cfg
        dy = canon(dy)
        dx = new_act(N, Cc, H, W, dy.dtype, dy.device)
        fn = lib.mt_avgpool2_bwd if kind == 2 else lib.mt_avgpool3s2_bwd
        L.check(fn(_mt(dy.dtype), _ptr(dy), _ptr(dx), N, H, W, padc(Cc), _stream()), "mt_avgpool_bwd")
        return dx, None


class _Patch4s2Multi(torch.autograd.Function):
    """Patches of a 4x4 / stride 2 / zero-padding 1 convolution of several inputs (same channels, any even sizes) as ONE batch
    of 4x4 mini-images [sum N_i Ho_i Wo_i, C, 4, 4] (mt_patch4s2_fwd); backward = the adjoint per input."""

    @staticmethod
    def forward(ctx, *xs):
        lib = L.load()
        xs = [canon(x) for x in xs]
        Cc, dt, dev = xs[0].shape[1], xs[0].dtype, xs[0].device
        for x in xs:
            if x.shape[1] != Cc or x.dtype != dt or x.shape[2] % 2 or x.shape[3] % 2:
                raise RuntimeError(f"patch4s2: inputs must share channels / type and have even sizes, got {tuple(x.shape)}")
        counts = [x.shape[0] * (x.shape[2] // 2) * (x.shape[3] // 2) for x in xs]
        col = new_act(sum(counts), Cc, 4, 4, dt, dev)
        row = 16 * padc(Cc) * col.element_size()
        offs, off = [], 0
        for n in counts:
            offs.append(off)
            off += n
        for k in range(0, len(xs), 4):              # (one launch per group of up to four inputs: the scales of a layer)
            grp = list(range(k, min(k + 4, len(xs))))
            G = len(grp)
            src = (C.c_void_p * G)(*[xs[i].data_ptr() for i in grp])
            dst = (C.c_void_p * G)(*[col.data_ptr() + offs[i] * row for i in grp])
            Ns = (C.c_int * G)(*[xs[i].shape[0] for i in grp])
            Hs = (C.c_int * G)(*[xs[i].shape[2] for i in grp])
            Ws = (C.c_int * G)(*[xs[i].shape[3] for i in grp])
            L.check(lib.mt_patch4s2_multi(_mt(dt), 0, G, src, dst, Ns, Hs, Ws, padc(Cc), _stream()), "mt_patch4s2_multi")
        ctx.shapes = [tuple(x.shape) for x in xs]
        ctx.counts = counts
        return col

    @staticmethod
    def backward(ctx, dcol):
        lib = L.load()
        dcol = canon(dcol)
        Cc = dcol.shape[1]
        row = 16 * padc(Cc) * dcol.element_size()
        outs, offs, off = [], [], 0
        for (N, _, H, W), n, need in zip(ctx.shapes, ctx.counts, ctx.needs_input_grad):
            outs.append(new_act(N, Cc, H, W, dcol.dtype, dcol.device) if need else None)
            offs.append(off)
            off += n
        for k in range(0, len(outs), 4):
            grp = list(range(k, min(k + 4, len(outs))))
            G = len(grp)
            src = (C.c_void_p * G)(*[dcol.data_ptr() + offs[i] * row for i in grp])
            dst = (C.c_void_p * G)(*[None if outs[i] is None else outs[i].data_ptr() for i in grp])
            Ns = (C.c_int * G)(*[ctx.shapes[i][0] for i in grp])
            Hs = (C.c_int * G)(*[ctx.shapes[i][2] for i in grp])
            Ws = (C.c_int * G)(*[ctx.shapes[i][3] for i in grp])
            L.check(lib.mt_patch4s2_multi(_mt(dcol.dtype), 1, G, src, dst, Ns, Hs, Ws, padc(Cc), _stream()), "mt_patch4s2_multi")
        return tuple(outs)


def patch4s2_multi(xs):
    """[x_0, x_1, ...] (NCHW, same C, even H / W) -> the 4x4 mini-image batch of their 4x4 / stride 2 / pad 1 patches"""
    return _Patch4s2Multi.apply(*xs)


def split_pixels(y, shapes):
    """y [sum N_i H_i W_i, C, 1, 1] (one output pixel per mini-image) -> views [N_i, C, H_i, W_i] of the same memory"""
    outs = []
    parts = torch.split(y, [n * h * w for n, h, w in shapes], dim=0)
    for part, (n, h, w) in zip(parts, shapes):
        c = part.shape[1]
        outs.append(part.permute(0, 2, 3, 1).reshape(n, h, w, c).permute(0, 3, 1, 2))
    return outs


class _Upsample2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = canon(x)
        N, Cc, H, W = x.shape
        y = new_act(N, Cc, 2 * H, 2 * W, x.dtype, x.device)
        L.check(L.load().mt_upsample2_fwd(_mt(x.dtype), _ptr(x), _ptr(y), N, 2 * H, 2 * W, padc(Cc), _stream()),
                "mt_upsample2_fwd")
        ctx.shape = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W = ctx.shape
        dy = canon(dy)
        dx = new_act(N, Cc, H, W, dy.dtype, dy.device)
        L.check(L.load().mt_upsample2_bwd(_mt(dy.dtype), _ptr(dy), _ptr(dx), N, 2 * H, 2 * W, padc(Cc), _stream()),
                "mt_upsample2_bwd")
        return dx


def upsample2_nearest(x):
    """nn.Upsample(scale_factor=2, mode='nearest')"""
    return _Upsample2.apply(x)


def avg_pool2(x):
    """nn.AvgPool2d(kernel_size=2, stride=2)"""
    return _Pool.apply(x, 2)


def avg_pool3s2(x):
    """nn.AvgPool2d(3, stride=2, padding=1, count_include_pad=False)"""
    return _Pool.apply(x, 3)


class _Gap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = canon(x)
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
        L.check(L.load().mt_gap_fwd(_mt(x.dtype), _ptr(x), _ptr(y), N, H * W, Cc, padc(Cc), _stream()), "mt_gap_fwd")
        ctx.cfg = (tuple(x.shape), x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (N, Cc, H, W), dt = ctx.cfg
        dy = _f32c(dy)
        dx = new_act(N, Cc, H, W, dt, dy.device)
        L.check(L.load().mt_gap_bwd(_mt(dt), _ptr(dy), _ptr(dx), N, H * W, Cc, padc(Cc), _stream()), "mt_gap_bwd")
        return dx


def global_avg_pool(x):
    """nn.AdaptiveAvgPool2d(1) followed by flatten: [N,C,H,W] -> fp32 [N,C]"""
    return _Gap.apply(x)


class _CatClass(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, cls):
        img = canon(img)
        cls = _f32c(cls)
        N, Cc, H, W = img.shape
        D = cls.shape[1]
        out = new_act(N, Cc + D, H, W, img.dtype, img.device)
        L.check(L.load().mt_cat_class_planes(_mt(img.dtype), _ptr(img), _ptr(cls), _ptr(out), N, H * W, Cc, D,
                                             _stream()), "mt_cat_class_planes")
        ctx.cfg = (Cc, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        Cc, D = ctx.cfg
        dout = canon(dout)
        N, _, H, W = dout.shape
        dimg = new_act(N, Cc, H, W, dout.dtype, dout.device)
        L.check(L.load().mt_slice_channels(_mt(dout.dtype), _ptr(dout), _ptr(dimg), N, H * W, Cc + D, Cc, _stream()),
                "mt_slice_channels")
        return dimg, None


def cat_class_planes(img, cls):
    """torch.cat([img, cls[:, :, None, None].repeat(1, 1, H, W)], dim=1) (networks.py:138-140)"""
    return _CatClass.apply(img, cls)


class _CatBatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *ts):
        ts = [canon(t) for t in ts]
        _, Cc, H, W = ts[0].shape
        ns = [t.shape[0] for t in ts]
        out = new_act(sum(ns), Cc, H, W, ts[0].dtype, ts[0].device)
        sz = H * W * padc(Cc)
        flat = out.as_strided((sum(ns) * sz,), (1,))
        o = 0
        for t, n in zip(ts, ns):
            flat[o * sz:(o + n) * sz].copy_(t.as_strided((n * sz,), (1,)))  # raw block copy incl. zero pads
            o += n
        ctx.ns = ns
        return out

    @staticmethod
    def backward(ctx, dy):
        return tuple(torch.split(dy, ctx.ns, dim=0))


def cat_batch(ts):
    """torch.cat(ts, dim=0) that keeps the padded-NHWC layout (no re-layout pass)."""
    return _CatBatch.apply(*ts)


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def _gs(g):
    return _f32c(g).reshape(1)


class _BceConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t):
        x = canon(x)
        N, Cc, H, W = x.shape
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        L.check(L.load().mt_bce_const_fwd(_mt(x.dtype), _ptr(x), t, _ptr(loss), N * H * W, Cc, padc(Cc), _stream()),
                "mt_bce_const_fwd")
        ctx.t = t
        ctx.save_for_backward(x)
        return loss

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        N, Cc, H, W = x.shape
        dx = new_act(N, Cc, H, W, x.dtype, x.device)
        L.check(L.load().mt_bce_const_bwd(_mt(x.dtype), _ptr(x), ctx.t, _ptr(_gs(g)), _ptr(dx), N * H * W, Cc,
                                          padc(Cc), _stream()), "mt_bce_const_bwd")
        return dx, None


class _GanConst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mode, t):
        x = canon(x)
        N, Cc, H, W = x.shape
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        L.check(L.load().mt_gan_const_fwd(_mt(x.dtype), mode, _ptr(x), t, _ptr(loss), N * H * W, Cc, padc(Cc),
                                          _stream()), "mt_gan_const_fwd")
        ctx.mode, ctx.t = mode, t
        ctx.save_for_backward(x)
        return loss

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        N, Cc, H, W = x.shape
        dx = new_act(N, Cc, H, W, x.dtype, x.device)
        L.check(L.load().mt_gan_const_bwd(_mt(x.dtype), ctx.mode, _ptr(x), ctx.t, _ptr(_gs(g)), _ptr(dx), N * H * W, Cc,
                                          padc(Cc), _stream()), "mt_gan_const_bwd")
        return dx, None, None


def mse_const(x, target_is_real):
    """nn.MSELoss()(x, ones/zeros expanded) -- GANLoss 'lsgan' (loss.py:44-45, 58-63)"""
    return _GanConst.apply(x, L.GAN_LSGAN, 1.0 if target_is_real else 0.0)


def hinge_dis(x, is_real):
    """relu(1 - x).mean() for real logits, relu(1 + x).mean() for fake ones (adain_model.py:209-210)"""
    return _GanConst.apply(x, L.GAN_HINGE_D, 1.0 if is_real else 0.0)


def neg_mean(x):
    """-x.mean(): the generator's hinge term (adain_model.py:293-295)"""
    return _GanConst.apply(x, L.GAN_NEG_MEAN, 1.0)


def signed_mean(x, negative):
    """-x.mean() (negative) or x.mean(): GANLoss 'wgangp' (loss.py:53-57)"""
    return _GanConst.apply(x, L.GAN_NEG_MEAN, 1.0 if negative else 0.0)


class _SubMean(torch.autograd.Function):
    """y = x - mean(other) on logit maps (the relativistic-average GAN terms, adain_model.py:206-208): composed of
    the existing kernels -- mean via mt_gan_const_fwd(NEG_MEAN), the shift via mt_scale_shift_act, and in the
    backward pass d(other) = -sum(g)/count via mt_gan_const_bwd."""

    @staticmethod
    def forward(ctx, x, other):
        lib = L.load()
        x, other = canon(x), canon(other)
        N, Cc, H, W = x.shape
        No, Co, Ho, Wo = other.shape
        Cp = padc(Cc)
        m = torch.empty((), dtype=torch.float32, device=x.device)
        L.check(lib.mt_gan_const_fwd(_mt(other.dtype), L.GAN_NEG_MEAN, _ptr(other), 0.0, _ptr(m), No * Ho * Wo, Co,
                                     padc(Co), _stream()), "mt_gan_const_fwd")            # t = 0: +mean(other)
        scale = torch.ones((N, Cp), dtype=torch.float32, device=x.device)
        shift = (-m).reshape(1, 1).expand(N, Cp).contiguous()
        y = new_act(N, Cc, H, W, x.dtype, x.device)
        L.check(lib.mt_scale_shift_act(_mt(x.dtype), _ptr(x), _ptr(scale), _ptr(shift), None, _ptr(y), N, H * W, Cp,
                                       L.ACT_NONE, 0.0, _stream()), "mt_scale_shift_act")
        ctx.save_for_backward(other)
        ctx.xshape = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        (other,) = ctx.saved_tensors
        g = canon(g)
        N, Cc, H, W = ctx.xshape
        No, Co, Ho, Wo = other.shape
        dother = None
        if ctx.needs_input_grad[1]:
            mg = torch.empty((), dtype=torch.float32, device=g.device)
            L.check(lib.mt_gan_const_fwd(_mt(g.dtype), L.GAN_NEG_MEAN, _ptr(g), 0.0, _ptr(mg), N * H * W, Cc, padc(Cc),
                                         _stream()), "mt_gan_const_fwd")
            gs = (mg * (-float(N * H * W * Cc))).reshape(1).contiguous()            # -sum(g)
            dother = new_act(No, Co, Ho, Wo, other.dtype, other.device)
            L.check(lib.mt_gan_const_bwd(_mt(other.dtype), L.GAN_NEG_MEAN, _ptr(other), 0.0, _ptr(gs), _ptr(dother),
                                         No * Ho * Wo, Co, padc(Co), _stream()), "mt_gan_const_bwd")
        return (g if ctx.needs_input_grad[0] else None), dother


def sub_mean(x, other):
    """x - torch.mean(other)"""
    return _SubMean.apply(x, other)


def bce_logits_const(x, target_is_real):
    """nn.BCEWithLogitsLoss()(x, ones/zeros expanded) -- GANLoss 'vanilla' (loss.py:58-63)"""
    return _BceConst.apply(x, 1.0 if target_is_real else 0.0)


class _BceTarget(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t):
        x, t = _f32c(x), _f32c(t)
        if x.shape != t.shape:
            raise RuntimeError(f"bce: shape mismatch {tuple(x.shape)} vs {tuple(t.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        L.check(L.load().mt_bce_target_fwd(_ptr(x), _ptr(t), _ptr(loss), x.numel(), _stream()), "mt_bce_target_fwd")
        ctx.save_for_backward(x, t)
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        dx = torch.empty_like(x)
        L.check(L.load().mt_bce_target_bwd(_ptr(x), _ptr(t), _ptr(_gs(g)), _ptr(dx), x.numel(), _stream()),
                "mt_bce_target_bwd")
        return dx, None


def bce_logits(x, target):
    """nn.BCEWithLogitsLoss()(x, target) on fp32 [N, D] class logits"""
    return _BceTarget.apply(x, target)


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        if a.dim() == 4:
            a, b = canon(a), canon(b)
            n, cnt, mt = _numel_padded(a), a.numel(), _mt(a.dtype)
        else:
            a, b = _f32c(a), _f32c(b)
            n, cnt, mt = a.numel(), a.numel(), L.MT_F32
        if a.shape != b.shape:
            raise RuntimeError(f"l1: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        L.check(L.load().mt_l1_fwd(mt, _ptr(a), _ptr(b), _ptr(loss), n, cnt, _stream()), "mt_l1_fwd")
        ctx.cfg = (n, cnt, mt)
        ctx.save_for_backward(a, b)
        return loss

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        n, cnt, mt = ctx.cfg

        def like(t):
            return new_act(*t.shape, t.dtype, t.device) if t.dim() == 4 else torch.empty_like(t)
        da = like(a) if ctx.needs_input_grad[0] else None
        db = like(b) if ctx.needs_input_grad[1] else None
        L.check(L.load().mt_l1_bwd(mt, _ptr(a), _ptr(b), _ptr(_gs(g)), _ptr(da), _ptr(db), n, cnt, _stream()),
                "mt_l1_bwd")
        return da, db


def l1_loss(a, b):
    """nn.L1Loss()(a, b) (mean)"""
    return _L1.apply(a, b)


class _L2Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = canon(x)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        L.check(L.load().mt_l2mean_fwd(_mt(x.dtype), _ptr(x), _ptr(loss), _numel_padded(x), x.numel(), _stream()),
                "mt_l2mean_fwd")
        ctx.save_for_backward(x)
        return loss

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = new_act(*x.shape, x.dtype, x.device)
        L.check(L.load().mt_l2mean_bwd(_mt(x.dtype), _ptr(x), _ptr(_gs(g)), _ptr(dx), _numel_padded(x), x.numel(),
                                       _stream()), "mt_l2mean_bwd")
        return dx


def l2_mean(x):
    """torch.mean(torch.pow(x, 2)) (adain_model.py:396-399)"""
    return _L2Mean.apply(x)


class _Reparam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = _f32c(mu), _f32c(logvar), _f32c(eps)
        z = torch.empty_like(mu)
        L.check(L.load().mt_reparam_fwd(_ptr(mu), _ptr(logvar), _ptr(eps), _ptr(z), mu.numel(), _stream()),
                "mt_reparam_fwd")
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = _f32c(dz)
        dmu, dlv = torch.empty_like(dz), torch.empty_like(dz)
        L.check(L.load().mt_reparam_bwd(_ptr(logvar), _ptr(eps), _ptr(dz), _ptr(dmu), _ptr(dlv), dz.numel(),
                                        _stream()), "mt_reparam_bwd")
        return dmu, dlv, None


def reparameterize(mu, logvar, eps):
    """eps * exp(0.5*logvar) + mu (networks.py:130-135)"""
    return _Reparam.apply(mu, logvar, eps)


class _KL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = _f32c(mu), _f32c(logvar)
        kl = torch.empty((), dtype=torch.float32, device=mu.device)
        L.check(L.load().mt_kl_fwd(_ptr(mu), _ptr(logvar), _ptr(kl), mu.numel(), _stream()), "mt_kl_fwd")
        ctx.save_for_backward(mu, logvar)
        return kl

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
        L.check(L.load().mt_kl_bwd(_ptr(mu), _ptr(logvar), _ptr(_gs(g)), _ptr(dmu), _ptr(dlv), mu.numel(), _stream()),
                "mt_kl_bwd")
        return dmu, dlv


def kl_sum(mu, logvar):
    """-0.5 * sum(1 + logvar - mu^2 - exp(logvar)) (adain_model.py:313-314)"""
    return _KL.apply(mu, logvar)


class _LossSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, *terms):
        w, gid, Wb, Wr = spec
        n, G = len(terms), len(Wb)
        ts = [_f32c(t.detach()).reshape(()) for t in terms]
        dev = ts[0].device
        out = torch.empty((G + 2,), dtype=torch.float32, device=dev)
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        ctx.cw = (C.c_float * n)(*w)
        ctx.cg = (C.c_int * n)(*gid)
        ctx.cWb = (C.c_float * G)(*Wb)
        cWr = (C.c_float * G)(*Wr)
        L.check(L.load().mt_loss_sum_fwd(ptrs, ctx.cw, ctx.cg, n, ctx.cWb, cWr, G, _ptr(out), _stream()), "mt_loss_sum_fwd")
        ctx.n, ctx.G = n, G
        total, logged = out[G], out
        ctx.mark_non_differentiable(logged)
        return total, logged

    @staticmethod
    def backward(ctx, g, _unused=None):
        d = torch.empty((ctx.n,), dtype=torch.float32, device=g.device)
        L.check(L.load().mt_loss_sum_bwd(_ptr(_gs(g)), ctx.cw, ctx.cg, ctx.n, ctx.cWb, ctx.G, _ptr(d), _stream()),
                "mt_loss_sum_bwd")
        return (None,) + tuple(d[i] for i in range(ctx.n))


def loss_sum(groups):
    """The model's loss expression in ONE launch (and one for its backward) instead of a chain of scalar adds / muls.
    ``groups``: list of (name, [(scalar tensor, weight), ...], weight in the differentiated total, weight in the reported
    total).  -> (total [differentiable], {name: group value}, reported total); the group values and the reported total
    are views of one device buffer (nothing is synchronised)."""
    w, gid, Wb, Wr, terms, names = [], [], [], [], [], []
    for g, (name, items, wb, wr) in enumerate(groups):
        names.append(name)
        Wb.append(float(wb))
        Wr.append(float(wr))
        for t, wt in items:
            terms.append(t)
            w.append(float(wt))
            gid.append(g)
    total, logged = _LossSum.apply((w, gid, Wb, Wr), *terms)
    G = len(names)
    return total, {n: logged[i] for i, n in enumerate(names)}, logged[G + 1]


# --------------------------------------------------------------------------------------
# optimizer step
# --------------------------------------------------------------------------------------
def adam_multi(params, grads, exp_avgs, exp_avg_sqs, lr, beta1, beta2, eps, wd, step):
    """One fused launch updating every tensor of an optimizer (torch.optim.Adam semantics)."""
    if not params:
        return
    dev = params[0].device
    ptrs, sizes = [], []
    for p, g, m, v in zip(params, grads, exp_avgs, exp_avg_sqs):
        for t in (p, g, m, v):
            _need_hip(t)
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("adam_multi needs contiguous fp32 tensors")
        ptrs += [p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()]
        sizes.append(p.numel())
    tp = torch.tensor(ptrs, dtype=torch.int64).to(dev, non_blocking=False)
    ts = torch.tensor(sizes, dtype=torch.int64).to(dev, non_blocking=False)
    L.check(L.load().mt_adam_multi(_ptr(tp), _ptr(ts), len(sizes), max(sizes), lr, beta1, beta2, eps, wd, step,
                                   _stream()), "mt_adam_multi")
    bump_epoch(params)


@contextlib.contextmanager
def frozen(*modules):
    """Temporarily mark the parameters of ``modules`` as not requiring grad (their weight
    gradients are computed-and-discarded by the reference; skipping them is trajectory
    identical -- SURVEY.md Appendix C)."""
    ps = [p for m in modules for p in m.parameters() if p.requires_grad]
    for p in ps:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in ps:
            p.requires_grad_(True)
