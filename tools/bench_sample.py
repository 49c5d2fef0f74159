"""Inference throughput of the sampling path (sample.py -> AdaINModel.forward_random / forward_reference) at the
reference's deployment size 540x960 (sample.py:79-91): content encoder + decoder (+ style encoder for references).

    python tools/bench_sample.py [--batch_size 1] [--precision bf16] [--iters 20]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch_size", type=int, default=1)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--hip_graph", action="store_true", help="replay the forwards from captured hipGraphs")
    o = ap.parse_args()
    from masterthesis_amd import models
    dev = torch.device("cuda", 0)
    a = argparse.Namespace(mode="test", precision=o.precision, input_dim=3, dim=64, enc_norm="instance", num_domains=4,
                           latent_dim=8, up_type="transpose", dec_norm="layer", use_dropout=False, init_type="normal",
                           init_gain=0.02, resume=None, gpu_ids=[0], batch_size=o.batch_size, concat=False, reparam=False,
                           hip_graph=o.hip_graph)
    torch.manual_seed(0)
    M = models.AdaINModel(a)
    M.initialize()
    img = torch.rand(o.batch_size, 3, o.height, o.width, device=dev) * 2 - 1
    c = torch.eye(4, device=dev)[[1] * o.batch_size]
    z = M.get_z_random(o.batch_size, 8)
    for name, fn in (("forward_random", lambda: M.forward_random(img, z, c)),
                     ("forward_reference", lambda: M.forward_reference(img, img.flip(3), c))):
        with torch.no_grad():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(o.iters):
                fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / o.iters
        print(f"{name}{' [hipGraph]' if o.hip_graph else ''}: {dt * 1e3:.2f} ms per batch of {o.batch_size} at {o.height}x{o.width} ({o.precision}) = "
              f"{o.batch_size / dt:.1f} images/s")


if __name__ == "__main__":
    main()
