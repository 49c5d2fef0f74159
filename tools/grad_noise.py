"""Round-off sensitivity of a step fixture: for each of the 7 optimizer steps of iteration 0, the relative L2 distance of
our fp32 gradients and of the reference's recorded fp32 gradients from the float64 oracle (per tensor for the first
generator phase).  Used to size the tolerances of tests/test_step_gpu.py (e.g. the --dis_sn fixture, whose generator
gradients jump by ~1.5 % per flipped LeakyReLU of the 4-channel discriminator).

    python tools/grad_noise.py adain_step_sn adain_step_lsgan
"""
import sys, os, tempfile, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_step_gpu as T
from helpers import sub
dev = torch.device("cuda", 0)
for name in sys.argv[1:]:
    class P:  # tmp_path stand-in
        def __init__(s): s.d = tempfile.mkdtemp()
        def __str__(s): return s.d
    z, meta, M, misc = T._build(name, P(), "fp32")
    truth = T._fp64_truth(z, meta, 1)
    batch = sub(z, "batch")
    src = misc.ReplaySource([z[f"rng/0/{i}"] for i in range(meta["rng_counts"][0])])
    misc.set_random_source(src)
    seen = []
    for net, opt in M.optimizer.items():
        opt._orig_step = opt.step
        def hooked(closure=None, _net=net, _opt=opt):
            torch.cuda.synchronize()
            seen.append((_net, {k: p.grad.detach().clone().cpu() for k, p in M.model[_net].named_parameters()}))
            return _opt._orig_step()
        opt.step = hooked
    M.update_lr(); M.set_inputs(batch); M.optimize_parameters(0)
    misc.set_random_source(None)
    t_loss, t_seen, t_state = truth[0]
    for j, ((net, g), (_, tg)) in enumerate(zip(seen, t_seen)):
        keys = [k for k in tg]
        a = torch.cat([g[k].double().flatten() for k in keys]); b = torch.cat([tg[k].double().flatten() for k in keys])
        r = torch.cat([torch.as_tensor(z[f"grad/0/{j}/{net}/{k}"]).double().flatten() for k in keys])
        print(f"{name} step{j} {net}: ours vs fp64 {((a-b).norm()/b.norm()).item():.3e}  reference-fp32 vs fp64 {((r-b).norm()/b.norm()).item():.3e}")
        if j == 2:
            for k in keys:
                e = ((g[k].double()-tg[k].double()).norm()/ (tg[k].double().norm()+1e-30)).item()
                er = ((torch.as_tensor(z[f"grad/0/{j}/{net}/{k}"]).double()-tg[k].double()).norm()/ (tg[k].double().norm()+1e-30)).item()
                print(f"      {k:40s} ours {e:.3e} ref {er:.3e} |g| {tg[k].norm().item():.3e}")
