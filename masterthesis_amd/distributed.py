"""Data parallelism for the step: one process per GPU, gradient exchange over RCCL/xGMI.

The reference's only multi-GPU hook is single-process ``nn.DataParallel`` (functions.py:98-101).  Here
every rank owns ``batch_size`` pairs and a full replica; after each of the four backward phases the
flat gradient buffer of every network that is about to be stepped is all-reduced (mean) on a side
HIP stream, bucket by bucket, so the exchange of network k overlaps the Adam launch of network k-1
and, for the discriminator phases, the generator work that follows.  SURVEY.md section 8(e).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not dist.is_initialized():
        if backend is None:
            # "nccl" is RCCL on ROCm.  MT_DIST_BACKEND=gloo: host-staged collectives on device tensors -- lets several
            # ranks share ONE GPU, which RCCL refuses (rehearsal of the N > 1 path on a single-GPU box)
            backend = os.environ.get("MT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
            # LOCAL_WORLD_SIZE only when the launcher set it (torchrun does): defaulting to the world size would switch the
            # kernel off on every multi-node run started without it
            nlocal = int(os.environ.get("LOCAL_WORLD_SIZE", "0"))
            if nlocal > torch.cuda.device_count():
                # several ranks on ONE GPU (rehearsals only): kernels that wait for sibling workgroups (the one-pass norm
                # backward) assume that no OTHER such kernel runs on the device at the same time -- two processes' partial
                # sets can hold each other's compute units until the bounded wait gives up and poisons the output.  One
                # process per GPU (the deployment, and what RCCL requires) never gets there.
                from . import hip_ops
                hip_ops.set_norm_onepass(False)
                if rank == 0:
                    import sys
                    print(f"masterthesis_amd: {nlocal} ranks share {torch.cuda.device_count()} GPU(s): one-pass norm backward off",
                          file=sys.stderr)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def rank_and_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def seed_rank_streams(base=None, group=None):
    """Give every rank its OWN random streams: the content noise, the reparameterisation eps,
    z_random and the dropout masks are per-sample draws, so replicas that all start from torch's default seed would
    train on duplicated noise.  Called after the weights were broadcast, so initialisation stays identical.
    ``base`` defaults to the seed the process already has (``torch.manual_seed(s)`` before building the model keeps
    meaning "reproducible run"); rank r reseeds the CPU and the device generator with ``base + r`` (rank 0 is left
    exactly as it was, so a single-process run is unchanged).  Returns the seed in use."""
    import random
    rank, world = rank_and_world(group)
    base = int(torch.initial_seed() if base is None else base)
    if world <= 1:
        return base
    # everyone agrees on rank 0's base seed (ranks may have been started with different ones)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([base], dtype=torch.int64, device=dev)
    dist.broadcast(t, src=0, group=group)
    base = int(t.item())
    if rank == 0:
        return base
    seed = (base + rank) % (2 ** 63)
    torch.manual_seed(seed)                 # CPU generator (Philox seeds of the noise / dropout kernels) + all devices
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    random.seed(seed)
    return seed


class NativeComm:
    """The library's own RCCL communicator (``mt_comm_*`` in include/mt_api.h): rank 0 creates the 128-byte id, it
    travels to the other ranks over the already initialised torch.distributed group, every rank joins with its device."""

    def __init__(self, group=None):
        import ctypes as C
        from . import _lib as L
        self.C, self.L, self.lib = C, L, L.load()
        rank, world = rank_and_world(group)
        idbuf = C.create_string_buffer(128)
        if rank == 0:
            L.check(self.lib.mt_comm_unique_id(idbuf), "mt_comm_unique_id")
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.frombuffer(idbuf, dtype=torch.uint8).clone().to(dev)
        if world > 1:
            dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().tolist())
        self.handle = C.c_void_p()
        L.check(self.lib.mt_comm_init(C.byref(self.handle), rank, world, raw, torch.cuda.current_device()), "mt_comm_init")

    def allreduce_async(self, buf, producer_stream):
        h = self.lib.mt_comm_allreduce_async(self.handle, self.C.c_void_p(buf.data_ptr()), buf.numel(),
                                             self.C.c_void_p(producer_stream))
        if h < 0:
            raise RuntimeError(f"mt_comm_allreduce_async failed: {self.lib.mt_last_error().decode(errors='replace')}")
        return h

    def wait(self, h, consumer_stream):
        self.L.check(self.lib.mt_comm_wait(self.handle, h, self.C.c_void_p(consumer_stream)), "mt_comm_wait")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mt_comm_destroy(self.handle)
            self.handle = None


class GradReducer:
    """All-reduce (mean) of flat gradient buffers, asynchronously on a side stream when on GPU.

    ``reduce(buffers)`` enqueues one collective per buffer and returns handles; ``wait(handle)`` makes
    the current stream wait for that buffer only.  Works with any backend (gloo on CPU for tests).  On the GPU the
    collective is issued either through torch.distributed (backend "nccl" = RCCL; the default) or, with
    ``MT_COMM=native``, through the library's own RCCL communicator (``mt_comm_*``): same stream/event structure, one
    ``ncclAllReduce(avg)`` per buffer, no separate scaling kernel."""

    def __init__(self, group=None):
        # MT_FORCE_REDUCER=1 keeps the exchange path active at world_size 1 (exercises RCCL + the side stream
        # on a single GPU; the all-reduce is then an identity)
        force = os.environ.get("MT_FORCE_REDUCER", "0") == "1"
        self.enabled = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.group = group
        self.world = dist.get_world_size(group) if self.enabled else 1
        self.side = None
        self.native = None
        self.log = None                 # tests: list that receives ("reduce" | "wait", ...) records
        if self.enabled and torch.cuda.is_available() and os.environ.get("MT_COMM", "torch") == "native":
            self.native = NativeComm(group)
        if self.enabled and torch.cuda.is_available():
            # the collectives' kernels are resident while the backward pass runs: kernels that need all slices of an image
            # resident together (one-pass norm backward) size themselves for the compute units that are left
            from . import hip_ops
            hip_ops.set_onepass_reserve(int(os.environ.get("MT_OP_RESERVE_CUS", "64")))

    def reduce(self, buffers):
        handles = []
        if not self.enabled:
            return [None for _ in buffers]
        on_gpu = buffers and buffers[0].is_cuda
        if self.log is not None:
            self.log.append(("reduce", [int(b.numel()) for b in buffers]))
        if on_gpu and self.native is not None:
            cur = torch.cuda.current_stream().cuda_stream
            return [("native", self.native.allreduce_async(b, cur)) for b in buffers]
        if on_gpu:
            if self.side is None:
                self.side = torch.cuda.Stream()
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())      # grads were produced on the compute stream
            self.side.wait_event(ready)
            with torch.cuda.stream(self.side):
                for b in buffers:
                    b.mul_(1.0 / self.world)
                    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    b.record_stream(self.side)
                    handles.append(ev)
        else:
            for b in buffers:
                b.mul_(1.0 / self.world)
                handles.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return handles

    def wait(self, handle):
        if handle is None:
            return
        if self.log is not None:
            self.log.append(("wait",))
        if isinstance(handle, tuple) and handle[0] == "native":
            self.native.wait(handle[1], torch.cuda.current_stream().cuda_stream)
        elif isinstance(handle, torch.cuda.Event):
            torch.cuda.current_stream().wait_event(handle)
        else:
            handle.wait()


def broadcast_parameters(tensors, src=0, group=None):
    """Make every replica start from rank ``src``'s weights."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in tensors:
            dist.broadcast(t, src=src, group=group)
