"""Throughput of the real-image input pipeline (PairedDataset -> DataLoader) against what the
training step consumes: decode + bicubic resize to load_size + crop + flip + normalise per image, pinned-memory batches.
Builds a throw-away tree of JPEG files (960x540, the size of the reference's weather frames) and reports images/s for
several worker counts.  CPU only -- run anywhere:
    python tools/bench_loader.py [--workers 4 8 16] [--batch_size 8] [--images 96]
"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, nargs="+", default=[2, 4, 8])
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--images", type=int, default=96, help="files per domain")
    ap.add_argument("--batches", type=int, default=24)
    o = ap.parse_args()
    from PIL import Image
    from masterthesis_amd.dataset import PairedDataset
    root = tempfile.mkdtemp()
    rng = np.random.default_rng(0)
    for dom in ("cloud", "fog", "rain", "sun"):
        os.makedirs(os.path.join(root, "train", dom))
        for i in range(o.images):
            # smooth random field: compresses like a photograph, so the decode cost is realistic
            base = rng.integers(0, 256, (34, 60, 3), dtype=np.uint8)
            Image.fromarray(base).resize((960, 540), Image.BICUBIC).save(os.path.join(root, "train", dom, f"{i}.jpg"),
                                                                         quality=90)
    args = argparse.Namespace(dataroot=root, mode="train", select_domains=None, num_domains=4, load_size=286,
                              crop_size=256, no_flip=False)
    ds = PairedDataset(args)
    print(f"{len(ds)} pairs per epoch, batch_size {o.batch_size} pairs ({2 * o.batch_size} images per batch)")
    for w in o.workers:
        dl = torch.utils.data.DataLoader(ds, batch_size=o.batch_size, shuffle=False, num_workers=w, drop_last=True,
                                         pin_memory=torch.cuda.is_available(), persistent_workers=w > 0,
                                         prefetch_factor=4 if w > 0 else None)
        n, t0 = 0, None
        while n < o.batches + 2:
            for batch in dl:
                n += 1
                if n == 2:
                    t0 = time.perf_counter()          # (worker start-up and the first prefetch window are not counted)
                if n >= o.batches + 2:
                    break
        dt = time.perf_counter() - t0
        print(f"num_workers {w:2d}: {2 * o.batch_size * o.batches / dt:8.1f} images/s "
              f"({dt / o.batches * 1e3:.1f} ms per batch of {2 * o.batch_size})")


if __name__ == "__main__":
    main()
