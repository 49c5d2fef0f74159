"""Inference driver -- mirror of the reference's ``src/sample.py:15-228``: translate every image of a
folder into each target domain, with styles drawn at random (``forward_random``) or taken from reference
images (``forward_reference``); optional image grid / multi-style grid.  Images are resized to 540x960 like
the reference (sample.py:50).

Like the reference the networks stay in train mode (the reference never calls ``.eval()``, so the content
noise is active); pass ``--eval`` for noise-free sampling.

    python -m masterthesis_amd.sample --model AdaINModel --dataroot <dir> --resume model_X.ckpt \\
           --num_domains 4 --targets cloud fog --batch_size 1
"""
import os

import torch

from . import hip_ops as ops
from .arguments import TestArguments
from .utils import TimerBlock, save_image_grid

DOMAIN_MAP = ["cloud", "fog", "rain", "sun"]
IMG_EXT = (".jpg", ".jpeg", ".png", ".ppm", ".bmp")


def load_rgb(path, size=(540, 960)):
    """PIL image -> fp32 [3,H,W] in [-1,1] (Resize + ToTensor + Normalize(0.5, 0.5))."""
    import numpy as np
    from PIL import Image
    img = Image.open(path).convert("RGB").resize((size[1], size[0]), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
    return (t - 0.5) / 0.5


class ImageList(torch.utils.data.Dataset):
    def __init__(self, root, size=(540, 960)):
        self.files = sorted(os.path.join(root, f) for f in os.listdir(root) if f.lower().endswith(IMG_EXT))
        self.size = size

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        return load_rgb(self.files[i], self.size)


class Sampler:
    size = (540, 960)

    def load_dataset(self, args):
        with TimerBlock("Loading Dataset"):
            if not os.path.isdir(args.dataroot):
                raise NotImplementedError("video input (ZED .svo / OpenCV) is out of scope; pass an image folder")
            return torch.utils.data.DataLoader(ImageList(args.dataroot, self.size), batch_size=args.batch_size,
                                               num_workers=args.num_workers, drop_last=True)

    def load_model(self, args):
        with TimerBlock("Creating model") as block:
            model = args.model(args)
            block.log("Initialize model")
            model.initialize()
            if getattr(args, "eval", False):
                for net in model.model:
                    model.model[net].eval()
            return model, model.device

    def load_image(self, args, path, device):
        return load_rgb(path, self.size).unsqueeze(0).repeat(args.batch_size, 1, 1, 1).to(device)

    def load_target(self, args, trg, device):
        onehot = torch.zeros((args.batch_size, args.num_domains))
        onehot[:, int(trg)] = 1
        return onehot.to(device)

    @torch.no_grad()
    def sample_batch(self, args, model, batch, trg, ref=None, z_sr=None, device="cuda"):
        trg_t = self.load_target(args, trg, device)
        batch = batch.to(device)
        if ref is not None:
            imgs, rt, mem = model.forward_reference(batch, self.load_image(args, ref, device), trg_t)
        elif z_sr is not None:
            imgs, rt, mem = model.forward_random(batch, z_sr, trg_t)
        else:
            raise ValueError("One of ref or z_sr values has to be provided.")
        return ops.to_nchw_f32(imgs), rt, mem

    @torch.no_grad()
    def sample(self, args, model, dataloader, trgs=None, refs=None, device="cuda"):
        with TimerBlock("Running model"):
            trgs = list(range(args.num_domains)) if trgs is None else trgs
            if refs is not None:
                assert len(trgs) == len(refs), "target and reference should match the shape"
            for t, trg in enumerate(trgs):
                z_sr = model.get_z_random(args.batch_size, args.latent_dim)
                for i, batch in enumerate(dataloader):
                    ref = refs[t] if refs is not None else None
                    imgs, _, _ = self.sample_batch(args, model, batch, trg, ref, None if ref else z_sr, device)
                    for j in range(len(imgs)):
                        save_image_grid(imgs[j:j + 1].cpu() / 2 + 0.5,
                                        os.path.join(args.display_dir, str(trg), f"image{t}_{i}_{j}.jpg"))

    @torch.no_grad()
    def generate_image_grid(self, args, model, dataloader, refs=None, trgs=None, device="cuda"):
        exetimes, memory, cols = [], [], []
        z_sr = model.get_z_random(args.batch_size, args.latent_dim) if refs is None else None
        trgs = list(range(args.num_domains)) if trgs is None else trgs
        for batch in dataloader:
            rows = [batch.to(device)]
            for t, trg in enumerate(trgs):
                ref = refs[t] if refs is not None else None
                imgs, rt, mem = self.sample_batch(args, model, batch, trg, ref, z_sr, device)
                rows.append(imgs)
                exetimes.append(rt)
                memory.append(mem)
            cols.append(torch.cat(rows, dim=3))
        images = torch.cat(cols, dim=2)
        print(f"Avg execution time : {sum(exetimes) / len(exetimes)}, cuda memory usage: {sum(memory) / len(memory)}")
        save_image_grid(images.cpu() / 2 + 0.5, "./grid.png")

    @torch.no_grad()
    def generate_multiple_styles(self, args, model, image, trg, refs=None, n_samples=4, device="cuda"):
        images = [image.to(device)]
        n = len(refs) if refs is not None else n_samples
        for i in range(n):
            ref = refs[i] if refs is not None else None
            z_sr = None if ref else model.get_z_random(args.batch_size, args.latent_dim)
            images.append(self.sample_batch(args, model, image, trg, ref, z_sr, device)[0])
        save_image_grid(torch.cat(images, dim=3).cpu() / 2 + 0.5, "./grid.png")

    def run(self, argv=None):
        with TimerBlock("Starting sampling") as block:
            ta = TestArguments()
            ta.parser.add_argument("--eval", action="store_true", help="sample in eval mode (no content noise)")
            args = ta.parse(argv)
            model, device = self.load_model(args)
            dataloader = self.load_dataset(args)
            targets = None if args.targets is None else [DOMAIN_MAP.index(t) if t in DOMAIN_MAP else int(t)
                                                         for t in args.targets]
            if args.gen_grid:
                block.log("Generating image grid")
                self.generate_image_grid(args, model, dataloader, args.reference, targets, device)
            elif args.gen_style:
                block.log("Generating multiple style image grid")
                self.generate_multiple_styles(args, model, next(iter(dataloader)), targets[0], args.reference,
                                              device=device)
            else:
                block.log("Running sample")
                self.sample(args, model, dataloader, targets, args.reference, device)


if __name__ == "__main__":
    Sampler().run()
