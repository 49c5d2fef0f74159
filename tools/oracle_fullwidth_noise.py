"""How far is the fp32 CPU oracle from the fp64 one at FULL width (dim 64, 256x256, batch_size 1, 4 domains, --ms_dis with the
real 2048-channel multi-scale discriminators)?  Sets the gradient bounds of tests/test_fullwidth_step_gpu.py: the GPU step is
compared with the fp32 oracle there (the fp64 one takes minutes on the GPU box's host), so the fp32 oracle's own distance from
fp64 per optimizer step is the floor of that comparison.

    python tools/oracle_fullwidth_noise.py [--res 256] [--threads 8]

Prints, per optimizer step of iteration 0: rel-L2 / cosine / norm ratio of the fp32 oracle's gradient against the fp64 oracle's,
and the relative differences of the 13 loss scalars.  Test infrastructure only (imports oracle/)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--domains", type=int, default=4)
    o = ap.parse_args()
    if o.threads:
        torch.set_num_threads(o.threads)
    from fullwidth_common import build_params, make_batch, run_oracle
    params = build_params(o.domains, o.res, ms=True)
    batch = make_batch(o.domains, o.res)
    t0 = time.time()
    l32, g32, rec = run_oracle(params, batch, o.domains, o.res, True, torch.float32, None)
    t1 = time.time()
    print(f"fp32 oracle: {t1 - t0:.1f} s", flush=True)
    l64, g64, _ = run_oracle(params, batch, o.domains, o.res, True, torch.float64, rec)
    print(f"fp64 oracle: {time.time() - t1:.1f} s", flush=True)
    for k in l64:
        print(f"loss {k:12s} fp64 {l64[k]: .6e}  fp32 rel {abs(l32[k] - l64[k]) / max(abs(l64[k]), 1e-2):.2e}")
    for j, ((net, a), (_, b)) in enumerate(zip(g32, g64)):
        va = torch.cat([a[k].double().flatten() for k in b])
        vb = torch.cat([b[k].double().flatten() for k in b])
        rel = ((va - vb).norm() / vb.norm()).item()
        cos = (torch.dot(va, vb) / (va.norm() * vb.norm())).item()
        print(f"step {j} {net:16s} rel {rel:.3e} cos {cos:.6f} ratio {(va.norm() / vb.norm()).item():.4f} |g| {vb.norm().item():.3e}")


if __name__ == "__main__":
    main()
