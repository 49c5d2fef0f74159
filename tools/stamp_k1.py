"""In-kernel s_memtime stamps of the 256x256 gather-GEMM (diagnostic build: make -C masterthesis_amd/csrc STAMPS=1)."""
import sys, ctypes as C
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import os
from masterthesis_amd import _lib
if os.environ.get("MT_DIAG_LIB"):          # diagnostic builds of the library (never the product path)
    _lib.LIB_PATH = os.path.abspath(os.environ["MT_DIAG_LIB"])
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Ci, Co, H, W = [int(v) for v in os.environ.get("MT_STAMP_SHAPE", "256,256,64,64").split(",")]     # (3x3 stride 1, reflection padding)
x = ops.canon(torch.randn(N, Ci, H, W, device=dev))
w = torch.randn(Co, Ci, 3, 3, device=dev) * 0.05
if len(sys.argv) > 2 and sys.argv[2] == "dgrad":      # the last launch of the stamped kernel is then the data gradient
    x.requires_grad_(True)
    gy = None
    for _ in range(10):
        y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")
        if gy is None:
            gy = ops.canon(torch.randn_like(y.float())).detach()
        y.backward(gy)
        x.grad = None
else:
    pm = "zero" if len(sys.argv) > 2 and sys.argv[2] == "fwdzero" else "reflect"
    with torch.no_grad():
        for _ in range(20): y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode=pm)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(16 * 4096, dtype=np.uint64)
lib.mt_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
rc = lib.mt_debug_stamps(buf.ctypes.data, buf.nbytes)
nb = int(os.environ.get("MT_STAMP_BLOCKS", N * H * W // 256 * max(Co // 256, 1)))
NS = int(os.environ.get("MT_STAMP_COUNT", "4"))          # 4: conv_pipe_kernel.hip (MT_STAMPS); 6: conv_pipe_patch_kernel.hip (MT_PP_STAMPS)
s = buf.reshape(4096, 16)[:nb, :NS].astype(np.int64)
d = np.diff(s, axis=1)
names = {4: ["prologue", "mainloop", "epilogue"], 6: ["setup", "first copies + addresses", "mainloop", "drain", "epilogue"]}[NS]
print("rc", rc, "blocks", nb)
print("median cycles: " + "  ".join(f"{n} {int(v)}" for n, v in zip(names, np.median(d, axis=0))) + f"  total {int(np.median(s[:, -1] - s[:, 0]))}")
print("p90    cycles: " + "  ".join(f"{n} {int(v)}" for n, v in zip(names, np.percentile(d, 90, axis=0))))
print("span first start -> last end: %d cycles" % (s[:, -1].max() - s[:, 0].min()))
print("start spread: %d  end spread: %d" % (s[:, 0].max() - s[:, 0].min(), s[:, -1].max() - s[:, -1].min()))
full = buf.reshape(4096, 16)[:nb].astype(np.int64)
if NS == 6 and (full[:, 6] > 0).all():
    print("inside the prologue (cycles after stamp 1): addresses done %d, first data landed + barrier %d, virtual cells %d" % tuple(
        int(np.median(full[:, k] - full[:, 1])) for k in (6, 7, 8)))
for i in range(NS):
    print(f"stamp {i}: min {int(s[:, i].min() - s[:, 0].min())}  median {int(np.median(s[:, i]) - s[:, 0].min())}  max {int(s[:, i].max() - s[:, 0].min())}")
