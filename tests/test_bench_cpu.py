"""bench.py's bookkeeping (no GPU): the algorithmic FLOP count per step is SURVEY.md 8(d)'s
F_step(B, H, W) = B * (H*W / 65536) * 2.407 TFLOP (2.566 with --ms_dis), images/s = 2 * B * N_gpus / step time, and the
workload string names BASELINE.json's configuration."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _o(**kw):
    a = dict(batch_size=8, crop_size=256, num_domains=2, precision="bf16", ms_dis=False)
    a.update(kw)
    return argparse.Namespace(**a)


def test_step_numbers_follow_survey_8d():
    import bench
    n = bench.step_numbers(_o(), dt=0.4, steps=10, world=1)              # 40 ms per step
    assert n["ms_per_step"] == 40.0
    assert n["images_per_sec"] == 400.0                                   # 2 * 8 * 1 / 0.040
    assert n["step_tflop_algorithmic"] == 19.26                           # 8 * 2.407
    assert abs(n["step_frac_of_mfma_peak"] - 19.256 / 0.040 / 2500.0) < 1e-3
    n = bench.step_numbers(_o(ms_dis=True, batch_size=16, num_domains=4), dt=0.8, steps=10, world=1)
    assert n["step_tflop_algorithmic"] == 41.06                           # 16 * 2.566
    n = bench.step_numbers(_o(crop_size=512, batch_size=2), dt=0.5, steps=10, world=8)
    assert n["step_tflop_algorithmic"] == 19.26                           # 2 * 4 * 2.407: linear in H*W
    assert n["images_per_sec"] == 2 * 2 * 8 * 10 / 0.5                    # whole-job aggregate over the 8 ranks
    n = bench.step_numbers(_o(precision="fp32"), dt=2.0, steps=10, world=1)
    assert abs(n["step_frac_of_mfma_peak"] - 19.256 / 0.2 / 157.3) < 1e-3   # fp32 runs are priced against the fp32 peak
    # the executed figure leaves out the encoder forward that the discriminator update shares with phase 3
    n = bench.step_numbers(_o(), dt=0.4, steps=10, world=1)
    assert abs(n["step_tflop_executed"] - 19.256 * (1 - (178.88 + 40.0) / 4 / 1203.47)) < 0.01
    assert n["step_frac_of_mfma_peak_executed"] < n["step_frac_of_mfma_peak"]


def test_workload_names_the_baseline_configuration():
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    w = bench.workload_name(_o())
    assert "AdaINModel" in w and "2 domains" in w and "256x256" in w and "batch_size 8" in w and "bf16" in w
    assert "AdaINModel" in base["configs"][1] and "256" in base["configs"][1] and "batch 8" in base["configs"][1]
    a = bench.model_args(_o(), "/tmp")
    # the reference's training defaults (arguments.py:85-118) that the benchmark step runs with
    assert (a.lr, a.wd, a.beta1, a.beta2, a.lambda_rec, a.lambda_cls, a.lambda_cls_G, a.d_iter, a.dim, a.latent_dim) == \
        (1e-4, 1e-4, 0.5, 0.999, 10.0, 1.0, 5.0, 3, 64, 8)
    assert a.gan_mode == "vanilla" and a.up_type == "transpose" and a.enc_norm == "instance" and a.dec_norm == "layer"
