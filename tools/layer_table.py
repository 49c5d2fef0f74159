"""Per-layer table of the convolution calls of one AdaINModel step (HIP events around every C-ABI conv call):
calls, average time, algorithmic TFLOP/s, and the roofline bound max(MFMA time, HBM time) for that shape.

    python tools/layer_table.py [--steps 3]
"""
import argparse
import collections
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

FIELDS = ["dtype", "transposed", "N", "H", "W", "Ci", "Co", "kh", "kw", "stride", "pad", "pad_mode", "out_pad", "act",
          "slope"]
PEAK_TF, PEAK_TB = 2500.0, 8.0


def padc(c):
    return (c + 7) // 8 * 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--crop_size", type=int, default=256)
    ap.add_argument("--ms_dis", action="store_true")
    a = ap.parse_args()
    o = argparse.Namespace(precision="bf16", num_domains=2, batch_size=a.batch_size, crop_size=a.crop_size, ms_dis=a.ms_dis)
    from masterthesis_amd import hip_ops as ops, models
    from masterthesis_amd.dataset import SyntheticDataset
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
    for it in range(2):
        step(it)
    torch.cuda.synchronize()
    ops.oplog_start()
    for it in range(2, 2 + a.steps):
        step(it)
    ev = ops.oplog_stop()
    agg = collections.OrderedDict()
    for kind, d, ms in ev:
        # (a grouped weight-gradient launch carries its group size behind the descriptor: G calls of that layer)
        g = d[16] if kind == "wgrad" and len(d) > 16 else 1
        e = agg.setdefault((kind, d[:15] if kind != "wgrad_rows" else d), [0, 0.0])
        e[0] += g
        e[1] += ms
    rows = []
    for (kind, d), (n, ms) in agg.items():
        f = dict(zip(FIELDS, d))
        if f["transposed"]:
            Ho = (f["H"] - 1) * f["stride"] - 2 * f["pad"] + f["kh"] + f["out_pad"]
            Wo = (f["W"] - 1) * f["stride"] - 2 * f["pad"] + f["kw"] + f["out_pad"]
            macs = f["N"] * f["H"] * f["W"] * f["Ci"] * f["Co"] * f["kh"] * f["kw"]
        else:
            Ho = (f["H"] + 2 * f["pad"] - f["kh"]) // f["stride"] + 1
            Wo = (f["W"] + 2 * f["pad"] - f["kw"]) // f["stride"] + 1
            macs = f["N"] * Ho * Wo * f["Ci"] * f["Co"] * f["kh"] * f["kw"]
        if kind == "wgrad_rows":                       # one shared launch of the row walker: d[16] layers, d[17] MACs in total
            macs = d[17]
        flop = 2.0 * macs
        xb = f["N"] * f["H"] * f["W"] * padc(f["Ci"]) * 2
        yb = f["N"] * Ho * Wo * padc(f["Co"]) * 2
        wb = f["Ci"] * f["Co"] * f["kh"] * f["kw"] * (2 if kind != "wgrad" else 4)
        byts = xb + yb + wb
        t_bound = max(flop / (PEAK_TF * 1e12), byts / (PEAK_TB * 1e12)) * 1e6
        avg = ms / n * 1e3
        name = f'{"convT" if f["transposed"] else "conv"} {f["kh"]}x{f["kw"]} s{f["stride"]} {f["Ci"]}->{f["Co"]} @{f["H"]}x{f["W"]} N{f["N"]}' \
               f'{" refl" if f["pad_mode"] else ""}'
        if kind == "wgrad_rows":
            name = f"{d[16]} 3x3 layers of a pass, first: {name}"[:44]
            t_bound = flop / (PEAK_TF * 1e12) * 1e6
        rows.append((ms / a.steps, kind[:6], name, max(n // a.steps, 1), avg, flop / avg / 1e6, t_bound, t_bound / avg))
    rows.sort(key=lambda r: -r[0])
    tot = sum(r[0] for r in rows)
    print(f"{'ms/step':>8s} {'kind':6s} {'layer':44s} {'calls':>5s} {'avg_us':>8s} {'TF/s':>7s} {'bound_us':>8s} {'frac':>5s}")
    for r in rows:
        print(f"{r[0]:8.3f} {r[1]:6s} {r[2]:44s} {r[3]:5d} {r[4]:8.1f} {r[5]:7.0f} {r[6]:8.1f} {r[7]:5.2f}")
    print(f"{tot:8.3f} total conv-call time per step (includes pack-free fwd, dgrad incl. fold, wgrad incl. unpack/colsum)")


if __name__ == "__main__":
    main()
