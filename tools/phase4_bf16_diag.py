"""Is the bf16 phase-4 gradient direction stable once every reduction is fixed-order?  

For each step fixture: the product step (iteration 0, recorded draws replayed) R times in bf16 and once in fp32, all in
deterministic mode (and optionally R times in bf16 with the atomics epilogue), whole-network gradient vectors of the two
phase-4 optimizer steps (content encoder, decoder) compared run against run and bf16 against fp32:
    python tools/phase4_bf16_diag.py [--runs 3] [--atomics] [fixture ...]
Prints one line per (fixture, network): cos / norm ratio of every bf16 run against the fp32 run, and the largest
element difference between bf16 runs (0 = bit-identical)."""
import argparse
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import load_gold, product_args, sub          # noqa: E402

FIXTURES = ["adain_step_d2", "adain_step_d4_b2", "base_step_concat_reparam", "adain_step_lsgan", "adain_step_hinge",
            "adain_step_ragan", "adain_step_nearest", "adain_step_sn", "adain_step_dc", "base_step_concat",
            "adain_step_dropout", "base_step_concat_dropout", "adain_step_norms", "adain_step_bn", "adain_step_ms",
            "adain_step_wgangp"]


def run_once(name, precision):
    """-> list of (network, flat fp64 gradient vector) per optimizer step of iteration 0."""
    from masterthesis_amd import models
    from masterthesis_amd.models.core import misc
    z, meta = load_gold(name)
    tmp = tempfile.mkdtemp()
    args = product_args(meta["args"], tmp, precision)
    M = getattr(models, meta["model"])(args)
    M.initialize()
    for net in M.model:
        M.model[net].load_state_dict(sub(z, f"init/{net}"))
    seen = []
    for net, opt in M.optimizer.items():
        orig = opt.step

        def hooked(closure=None, _net=net, _orig=orig):
            torch.cuda.synchronize()
            named = [(k, p.grad.detach().double().flatten().cpu()) for k, p in M.model[_net].named_parameters()
                     if p.grad is not None]
            seen.append((_net, torch.cat([v for _, v in named]), named))
            return _orig()
        opt.step = hooked
    misc.set_random_source(misc.ReplaySource([z[f"rng/0/{i}"] for i in range(meta["rng_counts"][0])]))
    try:
        M.update_lr()
        M.set_inputs(sub(z, "batch"))
        M.optimize_parameters(0)
    finally:
        misc.set_random_source(None)
    torch.cuda.synchronize()
    return seen


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fixtures", nargs="*", default=FIXTURES)
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--atomics", action="store_true", help="also R bf16 runs with the fused (atomic) statistics epilogue")
    ap.add_argument("--tensors", action="store_true", help="per-tensor cos / ratio of the phase-4 steps (first bf16 run)")
    o = ap.parse_args()
    from masterthesis_amd import hip_ops as ops
    for name in o.fixtures:
        modes = [("det", True)] + ([("atomics", False)] if o.atomics else [])
        for label, det in modes:
            ops.set_deterministic(det)
            try:
                ref = run_once(name, "fp32")
                runs = [run_once(name, "bf16") for _ in range(o.runs)]
            finally:
                ops.set_deterministic(False)
            for j, (net, v32, named32) in enumerate(ref):
                cs, rs, dmax = [], [], 0.0
                for r in runs:
                    v = r[j][1]
                    cs.append((torch.dot(v, v32) / (v.norm() * v32.norm() + 1e-300)).item())
                    rs.append((v.norm() / (v32.norm() + 1e-300)).item())
                    dmax = max(dmax, (v - runs[0][j][1]).abs().max().item())
                print(f"PH4DIAG {name:26s} {label:7s} step{j} {net:22s} cos " + " ".join(f"{c:+.3f}" for c in cs) +
                      "  ratio " + " ".join(f"{x:.3f}" for x in rs) + f"  run-to-run max diff {dmax:.2e}", flush=True)
                if o.tensors and j >= len(ref) - 2:
                    for (k, a), (_, b) in zip(runs[0][j][2], named32):
                        c = (torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)).item()
                        print(f"PH4TENS {name:22s} {label} step{j} {net}.{k:40s} cos {c:+.3f} ratio {(a.norm() / (b.norm() + 1e-300)).item():.3f} "
                              f"|fp32| {b.norm().item():.3e}", flush=True)


if __name__ == "__main__":
    main()
