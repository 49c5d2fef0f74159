import sys, os, tempfile
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, numpy as np
from helpers import load_gold, product_args, sub
from masterthesis_amd import models
from masterthesis_amd.models.core import misc
name, precision = sys.argv[1], sys.argv[2]
z, meta = load_gold(name)
args = product_args(meta["args"], tempfile.mkdtemp(), precision)
M = getattr(models, meta["model"])(args); M.initialize()
for net in M.model: M.model[net].load_state_dict(sub(z, f"init/{net}"))
batch = sub(z, "batch")
src = misc.ReplaySource([z[f"rng/0/{i}"] for i in range(meta["rng_counts"][0])]); misc.set_random_source(src)
seen = []
for net, opt in M.optimizer.items():
    opt._orig_step = opt.step
    def hooked(closure=None, _net=net, _opt=opt):
        torch.cuda.synchronize()
        seen.append((_net, {k: p.grad.detach().clone().cpu() for k, p in M.model[_net].named_parameters()}))
        return _opt._orig_step()
    opt.step = hooked
M.update_lr(); M.set_inputs(batch); M.optimize_parameters(0)
got = M.sync_losses()
for k, v in meta["losses"][0].items(): print(f"loss {k:12s} {got[k]:.6f} ref {v:.6f}")
from test_step_gpu import _fp64_truth
truth = _fp64_truth(z, meta, 1)[0][1]
for j, ((net, g), (_, tg)) in enumerate(zip(seen, truth)):
    for k, v in g.items():
        ref = tg[k].double(); v = v.double()
        r = ((v - ref).norm() / (ref.norm() + 1e-30)).item()
        cos = (v.flatten() @ ref.flatten() / (v.norm() * ref.norm() + 1e-30)).item()
        if r > float(sys.argv[3]): print(f"step{j} {net+'.'+k:50s} rel {r:.3e} cos {cos:.4f} refnorm {ref.norm():.3e} ours {v.norm():.3e}")
