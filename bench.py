#!/usr/bin/env python3
"""Benchmark of the north-star metric: train images/sec of the full G+D step of AdaINModel at 256x256
on synthetic batches, weak scaling over N GPUs (one process per GPU, RCCL gradient all-reduce).

Headline workload (run with the multi-scale PatchGAN that north_star names, single-scale alongside):
  N = 1   BASELINE.json configs[1]: 2 domains, 256x256, batch 8, bf16, --ms_dis
  N > 1   BASELINE.json configs[3]: 4 domains, 256x256, batch 16 per GPU, bf16, --ms_dis  (the DDP shape)
(`--single_scale`, `--batch_size`, `--num_domains` override; `extra` carries the single-scale step -- the headline of
rounds 1-2 --, configs[2], the configs[4] per-GPU shape at 512x512, fp32, batch 1 and the hipGraph replays.)

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = 2*B*N / step time (every step consumes B pairs = 2B images per
GPU).  `roofline` is measured live: HIP events around every launch of the dominant kernel (forward
implicit-GEMM of the 3x3 256->256 convolution on 64x64 maps) over the timed steps.
`cpu_baseline` times the CPU oracle (a port of the reference step to plain torch CPU ops) on a bounded
sample on this host's cores.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_STEP_TFLOP = {False: 2.407, True: 2.566}   # per B=1 at 256^2: single-scale D / --ms_dis
PEAK_BF16_TFLOPS = 2500.0                    # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def model_args(o, tmp):
    return argparse.Namespace(
        mode="train", precision=o.precision, logdir=tmp, checkpoint_dir=tmp, display_dir=tmp, input_dim=3,
        output_dim=3, dim=64, init_type="normal", init_gain=0.02, num_domains=o.num_domains, latent_dim=8,
        up_type="transpose", dec_norm="layer", enc_norm="instance", use_dropout=False, batch_size=o.batch_size,
        crop_size=o.crop_size, resume=None, resume_opt=None, gpu_ids=[0], dis_norm=None, dis_sn=False,
        ms_dis=o.ms_dis, num_scales=3, use_dis_content=False, lr=1e-4, wd=1e-4, beta1=0.5, beta2=0.999,
        lr_policy="step", n_iters=1000000, n_iter_decay=600000, last_iter=-1, d_iter=3, lambda_rec=10.0,
        lambda_cls=1.0, lambda_cls_G=5.0, gan_mode="vanilla", use_ragan=False, vgg_loss=None, concat=False,
        reparam=False, max_iter=1000000, synthetic_len=64, hip_graph=bool(getattr(o, "hip_graph", False)))


def cpu_baseline(o, args):
    """The CPU oracle (a port of the reference step to plain torch CPU ops, pinned against the reference by
    tests/golden) on this host's cores, as SURVEY.md section 8(d) specifies: batch_size 1 (2 images) at the bench
    resolution, fp32, 1 warm-up step + ``--cpu_steps`` timed steps (default 2); the thread count is stated.  A time guard
    keeps the default run bounded: if the warm-up step alone took more than ``--cpu_budget_s`` / 3 the timed steps are
    cut to one."""
    from oracle import step as ostep
    from masterthesis_amd.dataset import SyntheticDataset
    from masterthesis_amd.models.core import networks as N
    res = o.cpu_res or o.crop_size
    ms = bool(o.ms_dis and res >= 256)
    torch.manual_seed(0)
    nets = {"content_encoder": N.ContentEncoder(3, dim=64),
            "style_encoder": N.ReparameterizedStyleEncoder(3, output_dim=8, dim=64, num_domains=o.num_domains,
                                                           norm_layer=None, activation="lrelu"),
            "decoder": N.AdaINDecoder(3, dim=256, num_domains=o.num_domains, latent_dim=8)}
    for k in ("discriminator1", "discriminator2"):
        nets[k] = (N.MultiScaleDiscriminator(3, num_domains=o.num_domains) if ms else
                   N.Discriminator(3, dim=64, num_domains=o.num_domains, image_size=res))
    from masterthesis_amd.models.core.functions import init_weights
    params = {}
    for k, n in nets.items():
        init_weights(n, "normal", 0.02)
        params[k] = n.state_dict()
    oa = ostep.default_args(model="AdaINModel", dim=64, num_domains=o.num_domains, batch_size=1, crop_size=res, ms_dis=ms)
    O = ostep.OracleModel(params, oa)
    a2 = argparse.Namespace(crop_size=res, num_domains=o.num_domains, synthetic_len=4)
    ds = SyntheticDataset(a2, length=4, seed=99)
    cores = torch.get_num_threads()

    def step(it):
        batch = {k: v.unsqueeze(0) for k, v in ds[it % 4].items()}
        t0 = time.time()
        O.update_lr()
        O.set_inputs(batch)
        O.optimize_parameters(it)
        return time.time() - t0
    warm = step(0)
    n_timed = o.cpu_steps if warm * (1 + o.cpu_steps) <= o.cpu_budget_s else 1
    times = [step(1 + i) for i in range(n_timed)]
    dt = sum(times) / len(times)
    scale = (res * res) / float(o.crop_size * o.crop_size)
    note = "" if res == o.crop_size else f"; scaled by the pixel ratio ({res}^2/{o.crop_size}^2) to {o.crop_size}x{o.crop_size}"
    return {"value": round(2.0 / dt * scale, 5), "unit": "images/sec", "cores": cores, "kind": "port",
            "s_per_step": round(dt, 2), "warmup_step_s": round(warm, 2), "timed_steps": n_timed,
            "sample": f"CPU oracle (port of the reference step, torch CPU ops), full G+D step, batch_size 1 (2 images), "
                      f"{res}x{res}, fp32, {'multi' if ms else 'single'}-scale D, {cores} threads: 1 warm-up step "
                      f"({warm:.1f} s) + {n_timed} timed step(s), mean {dt:.1f} s{note}"}


def timed_run(o, rank, world, dev, steps, warmup, time_k1):
    """Build the model for ``o`` and time ``steps`` full G+D steps (after ``warmup`` untimed ones) between
    barrier + synchronize; -> (seconds [max over ranks], K1 launch timings, final losses)."""
    import torch.distributed as dist
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd import models
    from masterthesis_amd.dataset import SyntheticDataset
    tmp = tempfile.mkdtemp()
    args = model_args(o, tmp)
    torch.manual_seed(0)                    # identical initial weights; rank r then draws from seed 0 + r (Model.initialize)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):        # ("initialize network with ..." lines: stdout carries the JSON line only)
        M = models.AdaINModel(args)
        M.initialize()
    ds = SyntheticDataset(args, length=8, seed=1234 + rank)
    nb = 2
    batches = []
    for b in range(nb):     # synthetic batches, already resident in HBM before the timed region
        items = [ds[(b * o.batch_size + i) % 8] for i in range(o.batch_size)]
        batches.append({k: torch.stack([it[k] for it in items]).to(dev) for k in items[0]})

    def one_step(it):
        M.update_lr()
        M.set_inputs(batches[it % nb])
        M.optimize_parameters(it)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(warmup):
        one_step(it)

    def is_k1(d):
        return (not d.transposed and d.kh == 3 and d.stride == 1 and d.Ci == 256 and d.Co == 256
                and d.H == o.crop_size // 4)
    barrier()
    if time_k1:
        ops.kernel_timer_start(is_k1)
    t0 = time.time()
    for it in range(warmup, warmup + steps):
        one_step(it)
    barrier()
    dt = time.time() - t0
    k1_ms = ops.kernel_timer_stop() if time_k1 else []
    losses = M.sync_losses()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    del M, batches
    torch.cuda.empty_cache()
    return dt, k1_ms, losses


# share of the reference step's algorithmic FLOPs that this implementation does not execute: the content- and style-encoder
# forward of the discriminator update is shared with phase 3: ONE of the four content-encoder and ONE of the four
# style-encoder forward calls of a step (Ec fwd 178.88, Es fwd 40.00 GMAC over four calls each, of
# 1203.47 / 1283.05 GMAC) = 4.5 % / 4.3 %.  Every efficiency figure divides the FULL reference-equivalent F_step;
# the executed figure is reported next to it.
SHARED_ENCODER_GMAC = (178.88 + 40.00) / 4.0
F_STEP_GMAC = {False: 1203.47, True: 1283.05}


def step_numbers(o, dt, steps, world):
    ms_per_step = dt / steps * 1e3
    f_step = F_STEP_TFLOP[bool(o.ms_dis)] * o.batch_size * (o.crop_size ** 2 / 65536.0)
    shared = os.environ.get("MT_NO_ENCODER_SHARING", "0") != "1"
    f_exec = f_step * (1.0 - (SHARED_ENCODER_GMAC / F_STEP_GMAC[bool(o.ms_dis)] if shared else 0.0))
    peak = PEAK_BF16_TFLOPS if o.precision == "bf16" else PEAK_F32_TFLOPS
    return {"images_per_sec": round(2 * o.batch_size * world * steps / dt, 3), "ms_per_step": round(ms_per_step, 3),
            "step_tflop_algorithmic": round(f_step, 2),
            "step_tflop_executed": round(f_exec, 2),
            "step_tflops_achieved_per_gpu": round(f_step / (ms_per_step * 1e-3), 1),
            "step_tflops_executed_per_gpu": round(f_exec / (ms_per_step * 1e-3), 1),
            "step_frac_of_mfma_peak": round(f_step / (ms_per_step * 1e-3) / peak, 4),
            "step_frac_of_mfma_peak_executed": round(f_exec / (ms_per_step * 1e-3) / peak, 4)}


def hbm_kernels(o, rank, world, dev):
    """GB/s of the HBM-bound kernels inside the step (reported separately from the MFMA roofline): three more
    steps of the headline configuration with HIP events around every launch of the normalisation statistics / apply
    passes and of Adam; algorithmic bytes (operands read + written once) / launch time, against 8 TB/s."""
    from masterthesis_amd import hip_ops as ops
    import contextlib
    from masterthesis_amd import models
    from masterthesis_amd.dataset import SyntheticDataset
    args = model_args(o, tempfile.mkdtemp())
    torch.manual_seed(0)
    with contextlib.redirect_stdout(sys.stderr):
        M = models.AdaINModel(args)
        M.initialize()
    ds = SyntheticDataset(args, length=8, seed=1234 + rank)
    items = [ds[i % 8] for i in range(o.batch_size)]
    batch = {k: torch.stack([it[k] for it in items]).to(dev) for k in items[0]}
    for it in range(5):
        if it == 2:
            torch.cuda.synchronize()
            ops.hbm_timer_start()
        M.update_lr()
        M.set_inputs(batch)
        M.optimize_parameters(it)
    res = ops.hbm_timer_stop()
    del M
    torch.cuda.empty_cache()
    out = {}
    for name, (calls, ms, nbytes) in sorted(res.items()):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out[name] = {"calls_per_step": calls // 3, "ms_per_step": round(ms / 3, 3), "avg_us": round(ms / max(calls, 1) * 1e3, 1),
                     "algorithmic_GB_per_step": round(nbytes / 3 / 1e9, 3), "achieved_GBps": round(gbs, 1),
                     "frac_of_hbm_peak": round(gbs / 8000.0, 3)}
    return out


def comm_report(world, dev):
    """What the COMMUNICATOR says about the job (a SCALE run can check that RCCL really saw N ranks; `n_gpus` alone is the
    launcher's WORLD_SIZE): backend, world size as the process group reports it, the number of ranks counted by an all-reduce of
    ones over the very group the gradient exchange uses, the distinct devices behind them, and which exchange path
    (torch.distributed or the library's own mt_comm_*) carried the gradients."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_from_communicator": 1, "ranks_seen": 1, "devices_seen": 1,
                "exchange": "none (single process)"}
    ones = torch.ones(1, device=dev, dtype=torch.float32)
    dist.all_reduce(ones)
    # (rank, device) pairs: distinct physical devices behind the ranks (ranks sharing a GPU in a gloo rehearsal show up here)
    mine = torch.zeros(world, device=dev, dtype=torch.int64)
    import zlib
    if dev.type == "cuda":
        props = torch.cuda.get_device_properties(dev)
        name = str(getattr(props, "uuid", "")) or f"{os.uname().nodename}:{dev.index}"
    else:
        name = f"{os.uname().nodename}:cpu:{os.getpid()}"
    ident = zlib.crc32(name.encode()) + 1           # (crc32, not hash(): every rank must map one device to one number)
    mine[dist.get_rank()] = ident
    dist.all_reduce(mine)
    return {"backend": dist.get_backend(), "world_from_communicator": dist.get_world_size(),
            "ranks_seen": int(round(float(ones.item()))), "devices_seen": len(set(mine.tolist())),
            "exchange": "mt_comm (library RCCL communicator)" if os.environ.get("MT_COMM", "torch") == "native"
            else "torch.distributed"}


def workload_name(o):
    return (f"AdaINModel full G+D step, {o.num_domains} domains, {o.crop_size}x{o.crop_size}, batch_size {o.batch_size} "
            f"pairs/GPU ({2 * o.batch_size} images/GPU/step), {'multi-scale' if o.ms_dis else 'single-scale'} "
            f"discriminators, {o.precision}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch_size", type=int, default=0, help="pairs per GPU per step (default: 8 on one GPU = BASELINE "
                    "configs[1], 16 per GPU on several = configs[3])")
    ap.add_argument("--crop_size", type=int, default=256)
    ap.add_argument("--num_domains", type=int, default=0, help="default: 2 on one GPU (configs[1]), 4 on several (configs[3])")
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--ms_dis", action="store_true", help="(the default since round 3: multi-scale discriminators, the "
                    "PatchGAN north_star names; kept for old command lines)")
    ap.add_argument("--single_scale", action="store_true", help="single-scale discriminators (the reference's default and "
                    "the headline of rounds 1-2) for the headline run; reported under 'extra' otherwise")
    ap.add_argument("--no_hbm_kernels", action="store_true", help="skip the HBM-kernel GB/s block (three extra steps)")
    ap.add_argument("--cpu_res", type=int, default=0, help="resolution of the CPU-baseline steps (0 = the bench resolution)")
    ap.add_argument("--cpu_steps", type=int, default=2, help="timed CPU-baseline steps after one warm-up step")
    ap.add_argument("--cpu_budget_s", type=float, default=240.0)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--hip_graph", action="store_true", help="(the default on one GPU since round 4) replay the step from a "
                    "captured hipGraph; the dominant kernel is then timed live in a short eager leg of the same process")
    ap.add_argument("--eager", action="store_true", help="headline run with eager launches (rounds 1-3); always what runs on "
                    "several GPUs: the step is not captured while a gradient exchange is live")
    ap.add_argument("--no_extra", action="store_true", help="skip the additional configurations reported under 'extra'")
    o = ap.parse_args()

    from masterthesis_amd.distributed import init_from_env
    rank, world, local = init_from_env()
    if world != o.gpus:
        raise SystemExit(f"--gpus {o.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {o.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    local = local % torch.cuda.device_count()       # (MT_DIST_BACKEND=gloo rehearsals: several ranks on one GPU)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    o.ms_dis = not o.single_scale
    if o.batch_size <= 0:
        o.batch_size = 8 if world == 1 else 16
    if o.num_domains <= 0:
        o.num_domains = 2 if world == 1 else 4
    if o.hip_graph and world > 1:
        # the step is only captured without a live gradient exchange (translation._graph_mode: RCCL calls stay outside graphs
        # and no per-phase capture exists): a multi-GPU run would silently be an eager run labelled "hipGraph" -- refuse
        raise SystemExit("bench.py: --hip_graph is a single-GPU option (no graph capture with a live gradient exchange)")

    # Execution mode of the headline run.  One GPU: the whole optimize_parameters call -- ~1500 launches -- is captured once and
    # replayed (translation._optimize_graphed: same kernels, same work, the host's enqueue cost removed; "HIP streams and graphs
    # instead of a tracing compiler").  HIP events cannot bracket a kernel inside a replay, so the dominant kernel is timed live
    # in a short EAGER leg right after (same process, same shapes); the eager step time is reported under extra.eager.
    o.hip_graph = bool((o.hip_graph or not o.eager) and world == 1)
    if o.hip_graph:
        o.warmup = max(o.warmup, 6)
    dt, k1_ms, losses = timed_run(o, rank, world, dev, o.steps, o.warmup, time_k1=not o.hip_graph)
    eager_leg = None
    if o.hip_graph:
        import copy
        oe = copy.copy(o)
        oe.hip_graph = False
        ke = max(3, min(o.steps, 10))
        dte, k1_ms, _ = timed_run(oe, rank, world, dev, ke, 3, time_k1=True)
        eager_leg = dict(workload=workload_name(oe) + ", eager launches", steps=ke, warmup=3, **step_numbers(oe, dte, ke, world))
    # other configurations of BASELINE.json / SURVEY 8(d), measured in this same process right after the headline run
    # (single GPU only: the multi-GPU runs are the driver's scaling curve of the headline configuration)
    extra = {}
    if world == 1 and not o.no_extra:
        import copy
        variants = []
        if not o.ms_dis:
            variants.append(("ms_dis", dict(ms_dis=True)))
        else:
            variants.append(("single_scale", dict(ms_dis=False)))
        variants.append(("configs2_d4_b16_ms_dis", dict(num_domains=4, batch_size=16, ms_dis=True)))
        variants.append(("configs2_d4_b16_single_scale", dict(num_domains=4, batch_size=16, ms_dis=False)))
        # the per-GPU shape of BASELINE configs[4]: 4 domains, 512x512, multi-scale discriminators (large-activation stress);
        # batch 8 pairs = 16 images of 512x512 per step and GPU
        variants.append(("configs4_d4_512_b8_ms_dis", dict(num_domains=4, batch_size=8, crop_size=512, ms_dis=True)))
        variants.append(("fp32" if o.precision == "bf16" else "bf16",
                         dict(precision="fp32" if o.precision == "bf16" else "bf16")))
        variants.append(("batch_size_1", dict(batch_size=1)))       # the reference's scripts/train.sh batch size
        # the same step replayed from a captured hipGraph (--hip_graph): host enqueue cost removed
        variants.append(("batch_size_1_hip_graph", dict(batch_size=1, hip_graph=True)))
        if not o.hip_graph:
            variants.append(("hip_graph", dict(hip_graph=True)))
        for name, kw in variants:
            o2 = copy.copy(o)
            if name not in ("batch_size_1_hip_graph", "hip_graph"):
                o2.hip_graph = False            # the other configurations keep eager launches (comparable with rounds 1-3)
            for k, v in kw.items():
                setattr(o2, k, v)
            k = max(3, min(o.steps, 10))
            if getattr(o2, "crop_size", 256) > 256:
                k = min(k, 5)
            w = 6 if getattr(o2, "hip_graph", False) else 3      # 3 eager iterations + the capturing one come first
            dt2, _, _ = timed_run(o2, rank, world, dev, k, w, time_k1=False)
            extra[name] = dict(workload=workload_name(o2) + (", hipGraph replay" if getattr(o2, "hip_graph", False) else ""),
                               steps=k, warmup=w, **step_numbers(o2, dt2, k, world))
    hbm = None
    if world == 1 and not o.no_hbm_kernels:
        import copy
        oh = copy.copy(o)
        oh.hip_graph = False            # (HIP events around single launches: eager steps, like the roofline leg)
        hbm = hbm_kernels(oh, rank, world, dev)
    comm = comm_report(world, dev)
    if rank != 0:
        return
    N_img = 2 * o.batch_size
    k1_m = N_img * (o.crop_size // 4) ** 2
    nums = step_numbers(o, dt, o.steps, world)
    peak = PEAK_BF16_TFLOPS if o.precision == "bf16" else PEAK_F32_TFLOPS
    # dominant kernel: forward implicit GEMM, M = 2B*(H/4)^2 pixels, N = 256 couts, K = 9*256
    # (phase 4 runs the decoder on half batches: count each launch with its own pixel count)
    full = [t for t, n in k1_ms if n == N_img]
    k1_flop = 2.0 * k1_m * 256 * 2304
    k1_avg_ms = sum(full) / max(len(full), 1)
    tot_flop = sum(2.0 * n * (o.crop_size // 4) ** 2 * 256 * 2304 for _, n in k1_ms)
    tot_ms = sum(t for t, _ in k1_ms)
    achieved = tot_flop / (tot_ms * 1e-3) / 1e12 if k1_ms else 0.0
    # HBM-side traffic of the dominant kernel per launch is NOT measured in this run: it is copied from the committed
    # rocprofv3 PMC passes of this exact shape (FETCH_SIZE x2 correction + WRITE_SIZE, MI355X_MICROARCH.md; collected
    # with tools/pmc_k1.sh) and labelled as such
    traffic, mfma_busy, pmc_src, pmc_commit, pmc_kernel = None, None, None, None, None
    for cand in ("round4_k1_fwd_pmc.json",):
        # (only a PMC file collected on THIS round's binary is cited, with its commit inside)
        pmc_path = os.path.join(ROOT, "profiles", cand)
        if os.path.exists(pmc_path) and o.precision == "bf16" and o.batch_size == 8 and o.crop_size == 256:
            with open(pmc_path) as f:
                pmc = json.load(f)
            traffic, mfma_busy = pmc.get("traffic_bytes_per_launch"), pmc.get("mfma_busy_fraction")
            pmc_src = "profiles/" + cand
            pmc_commit, pmc_kernel = pmc.get("git_commit"), pmc.get("kernel")
            break
    out = {
        "metric": "train images/sec (G+D step), AdaINModel 256x256", "value": nums["images_per_sec"], "unit": "images/sec",
        "n_gpus": world, "steps": o.steps, "warmup": o.warmup, "ms_per_step": nums["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if o.precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": workload_name(o),
                   "global_batch_pairs": o.batch_size * world, "parallelism": f"dp{world}",
                   "baseline_config": "configs[1] + --ms_dis" if world == 1 else "configs[3]",
                   "step_tflop_algorithmic": nums["step_tflop_algorithmic"],
                   "step_tflop_executed": nums["step_tflop_executed"],
                   "step_tflops_achieved_per_gpu": nums["step_tflops_achieved_per_gpu"],
                   "step_tflops_executed_per_gpu": nums["step_tflops_executed_per_gpu"],
                   "step_frac_of_mfma_peak": nums["step_frac_of_mfma_peak"],
                   "step_frac_of_mfma_peak_executed": nums["step_frac_of_mfma_peak_executed"]},
        "roofline": {"bound": "mfma", "kernel": ("igemm_pipe_patch_kernel<bf16,nofold,taps9>" if o.precision == "bf16" else
                                                 "igemm_pipe_patch_kernel<f32,nofold,taps9>") + " fwd 3x3 s1 256->256 @64x64, "
                                                "256x256 tiles, patch-resident pixel operand (+ fused InstanceNorm-statistics "
                                                "epilogue)",
                     "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": pmc_src,
                     "traffic_git_commit": pmc_commit, "traffic_kernel": pmc_kernel,
                     "mfma_busy_fraction_pmc": mfma_busy,
                     "launches_timed": len(k1_ms), "avg_launch_ms_full_batch": round(k1_avg_ms, 4),
                     "flop_per_launch_full_batch": k1_flop},
        "final_losses": {k: round(float(v), 5) for k, v in losses.items()},
    }
    out["comm"] = comm
    if hbm:
        out["hbm_kernels"] = hbm
    out["execution"] = ("hipGraph replay of the whole optimize_parameters call (captured after 3 eager + 1 capturing iteration); "
                        "roofline kernel timed in an eager leg of the same process" if o.hip_graph else "eager launches")
    if eager_leg is not None:
        extra = dict(eager=eager_leg, **extra)
    if extra:
        out["extra"] = extra
    if world == 1 and not o.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(o, None)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
