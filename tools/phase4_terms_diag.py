"""Which loss term of phase 4 (backward_decoder_random: 10 * l1_recon_z + gan2 + gan2_cls) carries the bf16 / fp32
difference of a fixture?  Runs iteration 0 of the product step with the phase-4 loss reduced to ONE term (the group
weights of the other two set to zero inside ops.loss_sum) in fp32 and in bf16, deterministic mode, and compares
 * the gradient that reaches the decoder's output (d loss / d img_random, from a tensor hook on _translate's outputs),
 * the decoder's parameter gradient of the phase-4 optimizer step (whole vector, and the last layer alone).
    python tools/phase4_terms_diag.py adain_step_nearest [adain_step_d2 ...]"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import load_gold, product_args, sub          # noqa: E402


def run(name, precision, keep):
    from masterthesis_amd import hip_ops as ops, models
    from masterthesis_amd.models.core import misc
    z, meta = load_gold(name)
    args = product_args(meta["args"], tempfile.mkdtemp(), precision)
    M = getattr(models, meta["model"])(args)
    M.initialize()
    for net in M.model:
        M.model[net].load_state_dict(sub(z, f"init/{net}"))
    out = {"img_grad": [], "dec": None, "img": None}
    orig_loss_sum = ops.loss_sum

    def loss_sum(groups):
        if any(g[0] == "l1_recon_z" for g in groups) and keep != "all":
            groups = [(n, [(t, w if n == keep else 0.0) for t, w in items], wb, wr) for n, items, wb, wr in groups]
        return orig_loss_sum(groups)
    orig_translate = M._translate

    def translate(contents, styles, classes, per_call=None):
        res = orig_translate(contents, styles, classes, per_call=per_call)
        if per_call == 1 and torch.is_grad_enabled():
            out["img"] = [ops.to_nchw_f32(r).detach().double().cpu() for r in res]
            for r in res:
                r.register_hook(lambda g: out["img_grad"].append(ops.to_nchw_f32(g).detach().double().cpu()))
        return res
    M._translate = translate
    ops.loss_sum = loss_sum
    steps = []
    for net, opt in M.optimizer.items():
        orig = opt.step

        def hooked(closure=None, _net=net, _orig=orig):
            torch.cuda.synchronize()
            steps.append((_net, {k: p.grad.detach().double().cpu().clone() for k, p in M.model[_net].named_parameters()
                                 if p.grad is not None}))
            return _orig()
        opt.step = hooked
    misc.set_random_source(misc.ReplaySource([z[f"rng/0/{i}"] for i in range(meta["rng_counts"][0])]))
    ops.set_deterministic(True)
    try:
        M.update_lr()
        M.set_inputs(sub(z, "batch"))
        M.optimize_parameters(0)
    finally:
        misc.set_random_source(None)
        ops.set_deterministic(False)
        ops.loss_sum = orig_loss_sum
    torch.cuda.synchronize()
    out["dec"] = steps[-1][1]
    assert steps[-1][0] == "decoder"
    return out


def cos_ratio(a, b):
    a, b = a.flatten(), b.flatten()
    return (torch.dot(a, b) / (a.norm() * b.norm() + 1e-300)).item(), (a.norm() / (b.norm() + 1e-300)).item()


def main():
    for name in sys.argv[1:] or ["adain_step_nearest"]:
        for keep in ("all", "l1_recon_z", "gan2", "gan2_cls"):
            r32, r16 = run(name, "fp32", keep), run(name, "bf16", keep)
            g32, g16 = torch.cat([g.flatten() for g in r32["img_grad"]]), torch.cat([g.flatten() for g in r16["img_grad"]])
            i32, i16 = torch.cat([g.flatten() for g in r32["img"]]), torch.cat([g.flatten() for g in r16["img"]])
            c_img, r_img = cos_ratio(i16, i32)
            c_g, r_g = cos_ratio(g16, g32)
            v32 = torch.cat([v.flatten() for v in r32["dec"].values()])
            v16 = torch.cat([r16["dec"][k].flatten() for k in r32["dec"]])
            c_d, r_d = cos_ratio(v16, v32)
            last = list(r32["dec"])[-1] if "linear" not in list(r32["dec"])[-1] else [k for k in r32["dec"] if k.startswith("dec")][-1]
            c_l, r_l = cos_ratio(r16["dec"][last], r32["dec"][last])
            print(f"PH4TERM {name:22s} term {keep:11s} img_random cos {c_img:+.4f} ratio {r_img:.3f} | d/d img_random: |fp32| "
                  f"{g32.norm().item():.3e} mean {g32.mean().item():+.3e} cos {c_g:+.3f} ratio {r_g:.3f} | decoder grad cos {c_d:+.3f} "
                  f"ratio {r_d:.3f} | {last} cos {c_l:+.3f} ratio {r_l:.3f}", flush=True)


if __name__ == "__main__":
    main()
