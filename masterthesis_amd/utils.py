"""Helpers mirrored from the reference's ``src/utils.py``: AttributeDict (122-148), TimerBlock (150-173),
get_modules / module_to_dict (18-32), AverageMeter.  Image-grid / matplotlib helpers are visualisation
only and out of scope; ``save_image_grid`` writes a PPM/PNG via PIL when available."""
import os
import time
from collections import OrderedDict
from inspect import isclass


def get_modules(module, superclass=None, filter=None):
    names = [x for x in dir(module) if isclass(getattr(module, x))
             and (superclass is None or issubclass(getattr(module, x), superclass))]
    if filter:
        names = [m for m in names if filter in m]
    return names


def module_to_dict(module, exclude=()):
    return {x: getattr(module, x) for x in dir(module)
            if x not in exclude and isclass(getattr(module, x)) and getattr(module, x) not in exclude}


class AttributeDict(OrderedDict):
    """OrderedDict with attribute access; missing attributes read as None (reference utils.py:122-148)."""

    def __getattr__(self, attr):
        if attr.startswith("_OrderedDict__") or attr.startswith("__"):
            raise AttributeError(attr)
        return self.get(attr)

    def __setattr__(self, key, value):
        if key.startswith("_OrderedDict__"):
            return super().__setattr__(key, value)
        self[key] = value

    def __delattr__(self, item):
        del self[item]


class TimerBlock:
    """Context manager printing elapsed time; like the reference it measures process (CPU) time
    (utils.py:150-173) and additionally records wall time in ``wall``."""

    def __init__(self, title):
        print("{}".format(title))

    def __enter__(self):
        self.start = time.process_time()
        self._wall0 = time.time()
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.end = time.process_time()
        self.interval = self.end - self.start
        self.wall = time.time() - self._wall0
        self.log("Operation failed\n" if exc_type is not None else "Operation finished\n")

    def log(self, string):
        duration = time.process_time() - self.start
        units = "s"
        if duration > 60:
            duration, units = duration / 60.0, "m"
        print("  [{:.3f}{}] {}".format(duration, units, string), flush=True)


class AverageMeter:
    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)


def save_image_grid(tensor, path):
    """tensor: [N,3,H,W] in [0,1] (fp32, CPU) -> one image, samples stacked vertically (nrow=1)."""
    import numpy as np
    arr = (tensor.clamp(0, 1) * 255).byte().permute(0, 2, 3, 1).cpu().numpy()
    arr = np.concatenate(list(arr), axis=0)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    try:
        from PIL import Image
        Image.fromarray(arr).save(path)
    except Exception:  # PIL missing or unknown extension: raw PPM next to the requested name
        with open(os.path.splitext(path)[0] + ".ppm", "wb") as f:
            f.write(b"P6 %d %d 255\n" % (arr.shape[1], arr.shape[0]))
            f.write(arr.tobytes())
