"""Which torch (non-library) device kernels run inside a training step, and from where: torch.profiler over one step,
aggregated by (aten op, input shapes, innermost masterthesis_amd / autograd frame).

    python tools/torch_ops_in_step.py
"""
import argparse
import collections
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    from torch.profiler import ProfilerActivity, profile
    from masterthesis_amd import models
    from masterthesis_amd.dataset import SyntheticDataset
    o = argparse.Namespace(precision="bf16", num_domains=2, batch_size=8, crop_size=256, ms_dis=False)
    dev = torch.device("cuda", 0)
    args = bench.model_args(o, tempfile.mkdtemp())
    torch.manual_seed(0)
    M = models.AdaINModel(args)
    M.initialize()
    ds = SyntheticDataset(args, length=8, seed=1234)
    items = [ds[i] for i in range(o.batch_size)]
    batch = {k: torch.stack([it[k] for it in items]).to(dev) for k in items[0]}

    def step(it):
        M.update_lr()
        M.set_inputs(batch)
        M.optimize_parameters(it)
    for it in range(3):
        step(it)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step(3)
        torch.cuda.synchronize()
    rows = []
    for ev in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        us = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if us <= 0 or not ev.key.startswith("aten::"):
            continue
        frame = ""
        for f in ev.stack or []:
            if "masterthesis_amd" in f or "bench.py" in f:
                frame = f.split("masterthesis_amd/")[-1]
                break
        rows.append((us, ev.count, ev.key, str(ev.input_shapes)[:70], frame[:80]))
    rows.sort(key=lambda r: -r[0])
    print(f"torch device kernels in one step: {sum(r[1] for r in rows)} launches, {sum(r[0] for r in rows) / 1e3:.2f} ms")
    for us, n, name, shapes, frame in rows[:45]:
        print(f"{us:9.1f} us {n:4d}x {name:22s} {shapes:70s} {frame}")


if __name__ == "__main__":
    main()
