"""Datasets.  The batch contract of the reference's PairedDataset is kept: a dict with
``x1, x2`` (fp32 [3,H,W] in [-1,1]) and ``y1, y2`` (one-hot float [num_domains], two DIFFERENT domains)
(reference src/dataset.py:159-180).  ``SyntheticDataset`` is the benchmark / test input named by
BASELINE.json (synthetic 3xHxW batches); real-image loading mirrors SingleDataset/PairedDataset
(dataset.py:97-157) with PIL + torch only (torchvision is not available in this image)."""
import os
import random

import torch
from torch.utils.data import Dataset

IMG_EXTENSIONS = (".jpg", ".jpeg", ".png", ".ppm", ".bmp", ".tif", ".tiff", ".webp")


class SyntheticDataset(Dataset):
    """Deterministic synthetic pairs: U(-1,1) images, two distinct random domains per sample."""

    def __init__(self, args, return_paths=False, length=None, seed=None):
        self.args = args
        self.size = int(length if length is not None else getattr(args, "synthetic_len", 64))
        # the same dataset on every rank: the train loop's DistributedSampler hands each rank its own indices
        self.seed = int(seed if seed is not None else 1234)
        self.targets = list(range(args.num_domains))

    def __len__(self):
        return self.size

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(index))
        H = W = self.args.crop_size
        D = self.args.num_domains
        x1 = torch.rand(3, H, W, generator=g) * 2 - 1
        x2 = torch.rand(3, H, W, generator=g) * 2 - 1
        d1 = int(torch.randint(0, D, (1,), generator=g))
        d2 = (d1 + 1 + int(torch.randint(0, D - 1, (1,), generator=g))) % D if D > 1 else d1
        eye = torch.eye(D)
        return {"x1": x1, "x2": x2, "y1": eye[d1], "y2": eye[d2]}


def is_image_file(name):
    return name.lower().endswith(IMG_EXTENSIONS)


class SingleDataset(Dataset):
    """One image + one-hot domain label per item (reference dataset.py:97-157).
    Transform chain: bicubic resize to load_size -> random (train) / centre crop to crop_size ->
    random horizontal flip -> [0,1] -> normalise to [-1,1]."""

    def __init__(self, args, return_paths=False):
        self.args = args
        self.root = os.path.join(args.dataroot, args.mode)
        self.dataset, self.targets, self.target_names = self._make_dataset(self.root, args.select_domains)
        assert args.num_domains == len(self.targets)
        self.return_paths = return_paths
        self.size = max(map(len, self.dataset.values()))

    def _make_dataset(self, root, select_domains=None):
        if select_domains is not None:
            assert set(select_domains) <= set(os.listdir(root)), "Provided domain directories could not be found"
            domains = select_domains
        else:
            domains = os.listdir(root)
        dataset = {}
        for i, domain in enumerate(sorted(domains)):
            ddir = os.path.join(root, domain)
            dataset[i] = [os.path.join(ddir, f) for f in os.listdir(ddir) if is_image_file(f)]
        return dataset, sorted(dataset.keys()), domains

    def load_image(self, img_name, dim=3):
        import numpy as np
        from PIL import Image
        a = self.args
        img = Image.open(img_name).convert("RGB").resize((a.load_size, a.load_size), Image.BICUBIC)
        if a.mode == "train":
            top = random.randint(0, a.load_size - a.crop_size)
            left = random.randint(0, a.load_size - a.crop_size)
        else:
            top = left = (a.load_size - a.crop_size) // 2
        img = img.crop((left, top, left + a.crop_size, top + a.crop_size))
        t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        if not a.no_flip and random.random() < 0.5:
            t = t.flip(2)
        t = (t - 0.5) / 0.5
        if dim == 1:
            t = (t[0] * 0.299 + t[1] * 0.587 + t[2] * 0.114).unsqueeze(0)
        return t

    def get_onehot(self, index, shape):
        v = torch.zeros(shape)
        v[index] = 1
        return v

    def __len__(self):
        return self.size

    def __getitem__(self, index):
        y_src = random.choice(self.targets)
        x_src = self.dataset[y_src][index % len(self.dataset[y_src])]
        item = {"x": self.load_image(x_src), "y": self.get_onehot(y_src, (self.args.num_domains,))}
        if self.return_paths:
            item["x_path"] = x_src
        return item


class PairedDataset(SingleDataset):
    """Two images from two different domains (reference dataset.py:159-180)."""

    def __init__(self, args, return_paths=False):
        super().__init__(args, return_paths)
        if args.select_domains is not None:
            assert len(args.select_domains) >= 2

    def __getitem__(self, index):
        y1_src, y2_src = random.sample(self.targets, 2)
        x1_src = self.dataset[y1_src][index % len(self.dataset[y1_src])]
        x2_src = self.dataset[y2_src][index % len(self.dataset[y2_src])]
        item = {"x1": self.load_image(x1_src), "x2": self.load_image(x2_src),
                "y1": self.get_onehot(y1_src, (self.args.num_domains,)),
                "y2": self.get_onehot(y2_src, (self.args.num_domains,))}
        if self.return_paths:
            item.update(x1_path=x1_src, x2_path=x2_src)
        return item
