"""Multi-process (gloo, world_size 2) checks of the data-parallel path on CPU:
 * GradReducer averages flat gradient buffers across ranks (async handles, any backend);
 * sharding the batch over ranks + mean all-reduce + the KL world_size scaling reproduces the
   single-process gradient at the global batch, using the CPU oracle for the math."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker_reduce(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from masterthesis_amd.distributed import GradReducer, broadcast_parameters, init_from_env
    r, w, _ = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    red = GradReducer()
    assert red.enabled and red.world == world
    bufs = [torch.full((1000,), float(rank + 1)), torch.arange(10, dtype=torch.float32) * (rank + 1)]
    handles = red.reduce(bufs)
    for h in handles:
        red.wait(h)
    p = torch.full((4,), float(rank))
    broadcast_parameters([p], src=0)
    q.put((rank, bufs[0][0].item(), bufs[1].tolist(), p.tolist()))
    dist.destroy_process_group()


def test_grad_reducer_mean_allreduce_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_reduce, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, a, b, p in out:
        assert a == pytest.approx(1.5)
        assert b == pytest.approx([i * 1.5 for i in range(10)])
        assert p == [0.0] * 4


def _phase3_grads(O, rng, kl_scale):
    """gradient of the phase-3 generator loss for the oracle model O (KL term scaled by kl_scale)."""
    import torch.nn.functional as F
    from oracle import nets
    a, B, img, c = O.args, O.args.batch_size, O.img, O.c_org
    for n in ("content_encoder", "style_encoder", "decoder"):
        O.opt[n].zero_grad()
    cls_a, cls_b = torch.split(c, B)
    z_c = O.Ec(img, rng)
    z_ca, z_cb = torch.split(z_c, B)
    z_s, mu, logvar = O.Es(img, c, rng)
    z_sa, z_sb = torch.split(z_s, B)
    img_ba, img_aa = torch.split(O.Dec(torch.cat((z_cb, z_ca)), torch.cat((z_sa, z_sa)), torch.cat((cls_a, cls_a))), B)
    img_ab, img_bb = torch.split(O.Dec(torch.cat((z_ca, z_cb)), torch.cat((z_sb, z_sb)), torch.cat((cls_b, cls_b))), B)
    img_self = torch.cat((img_aa, img_bb))
    adv, cls = O._g_adv("discriminator1", torch.cat((img_ba, img_ab)), c)
    kl = torch.sum(1 + logvar - mu ** 2 - logvar.exp()) * -0.5 * 0.01
    loss = adv + cls + F.l1_loss(img, img_self) * a.lambda_rec + torch.mean(z_c ** 2) * 0.01 + kl * kl_scale
    loss.backward()
    return {n: torch.cat([p.grad.flatten() for p in O.P[n].values() if p.grad is not None])
            for n in ("content_encoder", "style_encoder", "decoder")}


def _make_oracle(batch_size):
    from helpers import load_gold, sub
    from oracle import step
    z, meta = load_gold("adain_step_d4_b2")
    a = meta["args"]
    args = step.default_args(**{k: a[k] for k in vars(step.default_args()) if k in a})
    args.model, args.batch_size = "AdaINModel", batch_size
    nets_present = sorted({k.split("/")[1] for k in z.files if k.startswith("init/")})
    O = step.OracleModel({n: sub(z, f"init/{n}") for n in nets_present}, args, dtype=torch.float64)
    return O, sub(z, "batch")


class _FixedRng:
    """per-sample deterministic draws so a shard sees exactly the rows of the global batch it owns"""

    def __init__(self, rows, total):
        self.rows, self.total, self.k = rows, total, 0

    def _draw(self, shape):
        g = torch.Generator().manual_seed(1000 + self.k)
        self.k += 1
        full = torch.randn((self.total,) + tuple(shape[1:]), generator=g, dtype=torch.float64)
        return full[self.rows]

    noise = eps = z = _draw


def _worker_shard(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from masterthesis_amd.distributed import GradReducer, init_from_env
    init_from_env(backend="gloo")
    red = GradReducer()
    O, batch = _make_oracle(batch_size=1)
    shard = {k: v[rank:rank + 1] for k, v in batch.items()}        # global batch_size 2 -> one pair per rank
    O.set_inputs(shard)
    # rows of the global [x1_0, x1_1, x2_0, x2_1] batch owned by this rank
    g = _phase3_grads(O, _FixedRng([rank, 2 + rank], 4), kl_scale=float(red.world))
    bufs = [g[n].clone() for n in ("content_encoder", "style_encoder", "decoder")]
    for h in red.reduce(bufs):
        red.wait(h)
    q.put((rank, [b.numpy() for b in bufs]))
    dist.destroy_process_group()


def test_sharded_gradients_equal_global_batch_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_shard, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process at the global batch (batch_size 2), KL unscaled
    O, batch = _make_oracle(batch_size=2)
    O.set_inputs(batch)
    ref = _phase3_grads(O, _FixedRng([0, 1, 2, 3], 4), kl_scale=1.0)
    for i, n in enumerate(("content_encoder", "style_encoder", "decoder")):
        for rank in range(world):
            got = torch.from_numpy(out[rank][i])
            rel = ((got - ref[n]).norm() / ref[n].norm()).item()
            assert rel < 1e-9, f"rank {rank} {n}: sharded+all-reduced gradient differs from global batch by {rel}"


def _worker_product_model(rank, world, port, q, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import argparse
    from helpers import load_gold, product_args
    from masterthesis_amd import dataset, models
    from masterthesis_amd.distributed import init_from_env
    from masterthesis_amd.train import Trainer
    init_from_env(backend="gloo")
    _, meta = load_gold("adain_step_sn")                # spectral norm: the u / v BUFFERS must be broadcast too
    args = product_args(meta["args"], os.path.join(tmp, str(rank)))
    args.dataset, args.num_workers, args.synthetic_len = dataset.SyntheticDataset, 0, 8
    torch.manual_seed(1000 * (rank + 1))                # deliberately different initial weights per rank
    M = models.AdaINModel(args)
    M.initialize()                                      # -> rank 0's weights everywhere, then per-rank random streams
    state = torch.cat([t.detach().flatten().double() for net in M.model for t in M.model[net].state_dict().values()])
    draws = (torch.randn(4).tolist(), int(torch.randint(0, 2 ** 62, (1,)).item()))
    loader = Trainer().load_dataset(args)
    first = next(iter(loader))
    q.put((rank, state.sum().item(), state.abs().sum().item(), M.rng_seed, draws, first["x1"].double().sum().item(),
           len(loader)))
    dist.destroy_process_group()


def test_product_model_two_ranks_same_weights_different_streams(tmp_path):
    """What the reference gets from one process feeding nn.DataParallel (functions.py:98-101), restated for one process
    per GPU: replicas start from identical weights and buffers, but every rank reads its own share of the dataset and
    draws its own noise / eps / z_random (ranks used to be seeded identically and read the same files)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_product_model, args=(r, world, port, q, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, s0, a0, seed0, d0, x0, n0), (_, s1, a1, seed1, d1, x1, n1) = out
    assert (s0, a0) == (s1, a1), "replicas do not start from the same weights / buffers"
    assert seed1 == seed0 + 1 and d0 != d1, "ranks draw from the same random stream"
    assert x0 != x1, "ranks read the same samples"
    assert n0 == n1 == 8 // 2 // 1                      # 8 items, 2 ranks, batch_size 1


class _FakeOpt:
    def __init__(self, name, buf, trace):
        self.name, self.buf, self.trace = name, buf, trace

    def flat_grad(self):
        return self.buf

    def step(self):
        self.trace.append(("step", self.name, self.buf.clone()))


def _worker_phase_order(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from types import SimpleNamespace
    from masterthesis_amd.distributed import GradReducer, init_from_env
    from masterthesis_amd.models.translation import TranslationModel
    init_from_env(backend="gloo")
    red = GradReducer()
    red.log = []
    trace = []
    names = ("content_encoder", "style_encoder", "decoder")
    # recorded flat gradients: rank r holds (r + 1) * base
    base = {n: torch.arange(1, 6, dtype=torch.float32) * (i + 1) for i, n in enumerate(names)}
    fake = SimpleNamespace(reducer=red, optimizer={n: _FakeOpt(n, base[n] * (rank + 1), trace) for n in names})
    fake._mark = lambda what: TranslationModel._mark(fake, what)
    TranslationModel._reduce_and_step(fake, names)
    # a deferred discriminator step (update_discriminator leaves it for the start of phase 4)
    d2 = _FakeOpt("discriminator2", torch.full((4,), float(rank + 1)), trace)
    fake._deferred_steps = [(d2, red.reduce([d2.flat_grad()]))]
    fake.__dict__["_deferred_steps"] = fake._deferred_steps
    TranslationModel._finish_deferred(fake)
    q.put((rank, [(k, n, b.tolist()) for k, n, b in trace], [e[0] for e in red.log]))
    dist.destroy_process_group()


def test_phase_exchange_and_step_order_two_ranks():
    """The product's ``_reduce_and_step`` / ``_finish_deferred`` on two gloo ranks with recorded flat gradient buffers:
    ALL buffers of a phase are handed to the exchange before the first optimizer steps, every optimizer steps on the
    rank average of ITS buffer, a deferred step waits for its own handle (reference hook replaced: functions.py:98-101)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_phase_order, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, trace, log in out:
        assert [n for _, n, _ in trace] == ["content_encoder", "style_encoder", "decoder", "discriminator2"]
        for i, (_, n, buf) in enumerate(trace[:3]):
            assert buf == pytest.approx([v * (i + 1) * 1.5 for v in range(1, 6)]), (rank, n, buf)     # mean of 1x and 2x
        assert trace[3][2] == pytest.approx([1.5] * 4)
        assert log[:2] == ["phase", "reduce"] and log.count("wait") == 4


def _worker_comm_report(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from masterthesis_amd.distributed import init_from_env
    init_from_env(backend="gloo")
    import bench
    q.put((rank, bench.comm_report(world, torch.device("cpu"))))
    dist.destroy_process_group()


def test_bench_comm_report_counts_the_ranks_of_the_communicator():
    """bench.py's `comm` block: world size and rank count come from the process group itself -- an all-reduce of
    ones -- not from the launcher's WORLD_SIZE, so a SCALE run can verify that the collective really spanned N ranks."""
    import bench
    assert bench.comm_report(1, torch.device("cpu"))["ranks_seen"] == 1          # no process group: single process
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_comm_report, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank in range(world):
        c = out[rank]
        assert c["backend"] == "gloo" and c["world_from_communicator"] == 2 and c["ranks_seen"] == 2
        assert c["devices_seen"] == 2 and c["exchange"] == "torch.distributed"
