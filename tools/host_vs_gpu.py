"""Is the step host-bound?  Host enqueue time of the timed steps (before the final synchronize) vs wall time."""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
o = argparse.Namespace(precision="bf16", num_domains=2, batch_size=int(os.environ.get("HB", "8")), crop_size=int(os.environ.get("HC", "256")), ms_dis=False)
from masterthesis_amd import models
from masterthesis_amd.dataset import SyntheticDataset
dev = torch.device("cuda", 0)
args = bench.model_args(o, tempfile.mkdtemp())
torch.manual_seed(0)
M = models.AdaINModel(args); M.initialize()
ds = SyntheticDataset(args, length=8, seed=1234)
items = [ds[i] for i in range(o.batch_size)]
batch = {k: torch.stack([it[k] for it in items]).to(dev) for k in items[0]}
def step(it):
    M.update_lr(); M.set_inputs(batch); M.optimize_parameters(it)
for it in range(3): step(it)
torch.cuda.synchronize()
t0 = time.time()
for it in range(3, 13): step(it)
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
print(f"host enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step")
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for it in range(13, 18): step(it)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
