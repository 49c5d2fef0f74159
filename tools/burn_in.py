"""Burn-in: N training steps of the benchmark configuration on synthetic data, eager or from the hipGraph, printing the
losses every `--every` steps and failing on the first non-finite value.  (Synthetic U(-1,1) images carry no structure:
the reconstruction losses fall towards the noise floor, the GAN terms hover around ln 2 -- the point is numerical health
over many steps with every fused path of the step in play.)
    python tools/burn_in.py --steps 1000 [--hip_graph] [--ms_dis] [--batch_size 8]"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--crop_size", type=int, default=256)
    ap.add_argument("--num_domains", type=int, default=2)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--ms_dis", action="store_true")
    ap.add_argument("--hip_graph", action="store_true")
    o = ap.parse_args()
    from masterthesis_amd import models
    from masterthesis_amd.dataset import SyntheticDataset
    dev = torch.device("cuda", 0)
    args = bench.model_args(o, tempfile.mkdtemp())
    torch.manual_seed(0)
    M = models.AdaINModel(args)
    M.initialize()
    ds = SyntheticDataset(args, length=64, seed=1)
    batches = []
    for b in range(64 // o.batch_size):
        items = [ds[b * o.batch_size + i] for i in range(o.batch_size)]
        batches.append({k: torch.stack([it[k] for it in items]).to(dev) for k in items[0]})
    t0 = time.time()
    for it in range(o.steps):
        M.update_lr()
        M.set_inputs(batches[it % len(batches)])
        M.optimize_parameters(it)
        if it % o.every == 0 or it == o.steps - 1:
            L = M.sync_losses()
            bad = [k for k, v in L.items() if not (v == v and abs(v) < 1e6)]
            print(f"it {it:5d} " + " ".join(f"{k} {v:.4f}" for k, v in L.items() if k in
                  ("d_total", "g_adv", "g_cls", "l1_self_rec", "l1_cc_rec", "l1_recon_z", "kl_zs", "total_g")), flush=True)
            if bad:
                raise SystemExit(f"non-finite losses at iteration {it}: {bad}")
    torch.cuda.synchronize()
    for n in M.model:
        assert all(torch.isfinite(p).all() for p in M.model[n].parameters()), n
    print(f"ok: {o.steps} steps in {time.time() - t0:.1f} s ({'hipGraph' if o.hip_graph else 'eager'})")


if __name__ == "__main__":
    main()
