#!/usr/bin/env python3
"""Benchmark of the north-star metric: train images/sec of the full G+D step of AdaINModel at 256x256
on synthetic batches (BASELINE.json configs[1]: 2 domains, batch 8, bf16, one MI355X), weak scaling
over N GPUs (one process per GPU, RCCL gradient all-reduce).

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
        reparam=False, max_iter=1000000, synthetic_len=64)


def cpu_baseline(o, args):
    """The CPU oracle (port of the reference step) on a bounded sample: ONE full G+D step at batch 1 and a
    reduced resolution, scaled by the pixel ratio (the step's work is linear in H*W)."""
    from oracle import step as ostep
    from masterthesis_amd.dataset import SyntheticDataset
    from masterthesis_amd.models.core import networks as N
    res = o.cpu_res
    torch.manual_seed(0)
    nets = {"content_encoder": N.ContentEncoder(3, dim=64),
            "style_encoder": N.ReparameterizedStyleEncoder(3, output_dim=8, dim=64, num_domains=o.num_domains,
                                                           norm_layer=None, activation="lrelu"),
            "decoder": N.AdaINDecoder(3, dim=256, num_domains=o.num_domains, latent_dim=8)}
    for k in ("discriminator1", "discriminator2"):
        nets[k] = (N.MultiScaleDiscriminator(3, num_domains=o.num_domains) if o.ms_dis and res >= 256 else
                   N.Discriminator(3, dim=64, num_domains=o.num_domains, image_size=res))
    from masterthesis_amd.models.core.functions import init_weights
    params = {}
    for k, n in nets.items():
        init_weights(n, "normal", 0.02)
        params[k] = n.state_dict()
    oa = ostep.default_args(model="AdaINModel", dim=64, num_domains=o.num_domains, batch_size=1, crop_size=res,
                            ms_dis=bool(o.ms_dis and res >= 256))
    O = ostep.OracleModel(params, oa)
    a2 = argparse.Namespace(crop_size=res, num_domains=o.num_domains, synthetic_len=1)
    item = SyntheticDataset(a2, length=1, seed=99)[0]
    batch = {k: v.unsqueeze(0) for k, v in item.items()}
    cores = torch.get_num_threads()
    t0 = time.time()
    O.update_lr()
    O.set_inputs(batch)
    O.optimize_parameters(0)
    dt = time.time() - t0
    scale = (res * res) / float(o.crop_size * o.crop_size)
    return {"value": 2.0 / dt * scale, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"1 full G+D step of the CPU oracle, batch_size 1 (2 images), {res}x{res}, fp32, "
                      f"{dt:.1f} s wall; scaled by the pixel ratio ({res}^2/{o.crop_size}^2) to {o.crop_size}x{o.crop_size}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch_size", type=int, default=8, help="pairs per GPU per step (BASELINE config 2: 8)")
    ap.add_argument("--crop_size", type=int, default=256)
    ap.add_argument("--num_domains", type=int, default=2)
    ap.add_argument("--precision", default="bf16", choices=["fp32", "bf16"])
    ap.add_argument("--ms_dis", action="store_true", help="multi-scale discriminators (default: single-scale, "
                    "the reference's default)")
    ap.add_argument("--cpu_res", type=int, default=128, help="resolution of the bounded CPU-baseline sample")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    o = ap.parse_args()

    from masterthesis_amd.distributed import init_from_env
    rank, world, local = init_from_env()
    if world != o.gpus:
        raise SystemExit(f"--gpus {o.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {o.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    from masterthesis_amd import hip_ops as ops
    from masterthesis_amd import models
    from masterthesis_amd.dataset import SyntheticDataset

    tmp = tempfile.mkdtemp()
    args = model_args(o, tmp)
    torch.manual_seed(0)
    M = models.AdaINModel(args)
    M.initialize()
    ds = SyntheticDataset(args, length=8, seed=1234 + rank)
    nb = 2
    batches = []
    for b in range(nb):     # synthetic batches, already resident in HBM before the timed region
        items = [ds[b * o.batch_size + i] for i in range(o.batch_size)]
        batches.append({k: torch.stack([it[k] for it in items]).to(dev) for k in items[0]})

    def one_step(it):
        M.update_lr()
        M.set_inputs(batches[it % nb])
        M.optimize_parameters(it)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(o.warmup):
        one_step(it)
    N_img = 2 * o.batch_size
    k1_m = N_img * (o.crop_size // 4) ** 2

    def is_k1(d):
        return (not d.transposed and d.kh == 3 and d.stride == 1 and d.Ci == 256 and d.Co == 256
                and d.H == o.crop_size // 4)
    barrier()
    ops.kernel_timer_start(is_k1)
    t0 = time.time()
    for it in range(o.warmup, o.warmup + o.steps):
        one_step(it)
    barrier()
    dt = time.time() - t0
    k1_ms = ops.kernel_timer_stop()
    losses = M.sync_losses()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        return
    ms_per_step = dt / o.steps * 1e3
    value = N_img * world * o.steps / dt
    f_step = F_STEP_TFLOP[bool(o.ms_dis)] * o.batch_size * (o.crop_size ** 2 / 65536.0)
    peak = PEAK_BF16_TFLOPS if o.precision == "bf16" else PEAK_F32_TFLOPS
    # dominant kernel: forward implicit GEMM, M = 2B*(H/4)^2 pixels, N = 256 couts, K = 9*256
    # (phase 4 runs the decoder on half batches: count each launch with its own pixel count)
    full = [t for t, n in k1_ms if n == N_img]
    k1_flop = 2.0 * k1_m * 256 * 2304
    k1_avg_ms = sum(full) / max(len(full), 1)
    tot_flop = sum(2.0 * n * (o.crop_size // 4) ** 2 * 256 * 2304 for _, n in k1_ms)
    tot_ms = sum(t for t, _ in k1_ms)
    achieved = tot_flop / (tot_ms * 1e-3) / 1e12 if k1_ms else 0.0
    # HBM-side traffic of the dominant kernel per launch: rocprofv3 PMC passes (FETCH_SIZE x2 correction +
    # WRITE_SIZE, MI355X_MICROARCH.md) collected with tools/bench_k1.py on this exact shape; see profiles/
    traffic, mfma_busy = None, None
    pmc_path = os.path.join(ROOT, "profiles", "round1_k1_fwd_pmc.json")
    if os.path.exists(pmc_path) and o.precision == "bf16" and o.batch_size == 8 and o.crop_size == 256:
        with open(pmc_path) as f:
            pmc = json.load(f)
        traffic, mfma_busy = pmc.get("traffic_bytes_per_launch"), pmc.get("mfma_busy_fraction")
    out = {
        "metric": "train images/sec (G+D step), AdaINModel 256x256", "value": round(value, 3), "unit": "images/sec",
        "n_gpus": world, "steps": o.steps, "warmup": o.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if o.precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": f"AdaINModel full G+D step, {o.num_domains} domains, {o.crop_size}x{o.crop_size}, "
                               f"batch_size {o.batch_size} pairs/GPU ({N_img} images/GPU/step), "
                               f"{'multi-scale' if o.ms_dis else 'single-scale'} discriminators, {o.precision}",
                   "global_batch_pairs": o.batch_size * world, "parallelism": f"dp{world}",
                   "step_tflop_algorithmic": round(f_step, 2),
                   "step_tflops_achieved_per_gpu": round(f_step / (ms_per_step * 1e-3), 1),
                   "step_frac_of_mfma_peak": round(f_step / (ms_per_step * 1e-3) / peak, 4)},
        "roofline": {"bound": "mfma", "kernel": ("igemm_pipe_kernel<bf16,256,256,512,4>" if o.precision == "bf16" else
                                                 "igemm_pipe_kernel<f32,256,256,512,4>") + " fwd 3x3 s1 256->256 @64x64 "
                                                "(+ fused InstanceNorm-statistics epilogue)",
                     "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": traffic,
                     "mfma_busy_fraction_pmc": mfma_busy,
                     "launches_timed": len(k1_ms), "avg_launch_ms_full_batch": round(k1_avg_ms, 4),
                     "flop_per_launch_full_batch": k1_flop},
        "final_losses": {k: round(float(v), 5) for k, v in losses.items()},
    }
    if world == 1 and not o.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(o, args)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
