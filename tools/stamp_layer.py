"""In-kernel s_memtime stamps of the 2-stage gather-GEMM on one layer (diagnostic build:
tools/diag_build.sh st2 -DMT_STAMPS2 conv_kernels.hip; run with MT_DIAG_LIB=_exp/libmt_st2.so).
usage: stamp_layer.py N Cin H Cout k stride [T for transposed]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from masterthesis_amd import _lib
if os.environ.get("MT_DIAG_LIB"):          # diagnostic builds of the library (never the product path)
    _lib.LIB_PATH = os.path.abspath(os.environ["MT_DIAG_LIB"])
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
N, Ci, H, Co, k, st = [int(a) for a in sys.argv[1:7]]
tr = len(sys.argv) > 7 and sys.argv[7] == "T"
x = ops.canon(torch.randn(N, Ci, H, H, device=dev))
if tr:
    w = torch.randn(Ci, Co, k, k, device=dev) * 0.05
    f = lambda: ops.conv_transpose2d(x, w, None, stride=st, pad=1, out_pad=1)
else:
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    f = lambda: ops.conv2d(x, w, None, stride=st, pad=k // 2, pad_mode="reflect")
with torch.no_grad():
    for _ in range(10): y = f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): y = f()
    e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
lib = C.CDLL(_lib.LIB_PATH)
buf = np.zeros(8 * 16384, dtype=np.uint64)
lib.mt_debug_stamps2.argtypes = [C.c_void_p, C.c_size_t]
rc = lib.mt_debug_stamps2(buf.ctypes.data, buf.nbytes)
s = buf.reshape(16384, 8)[:, :6].astype(np.int64)
nb = int((s[:, 0] > 0).sum())
s = s[:nb]
t0 = s[:, 0].min()
span = s[:, 5].max() - t0
print(f"layer N{N} {Ci}->{Co} @{H} k{k} s{st} {'T' if tr else ''}: {us:.1f} us per launch, {nb} blocks stamped, span {span} cycles -> {span/us/1e3:.2f} GHz-equivalent")
d = np.diff(s, axis=1)
names = ["setup(idx math)", "first k-step(load latency)", "rest of k loop", "epilogue issue", "store drain"]
names = ["setup", "issue0+kstep0", "k loop rest", "epilogue", "store drain"]
print("median cycles:", ", ".join(f"{n} {int(np.median(d[:, i]))}" for i, n in enumerate(names)), " total", int(np.median(s[:, 5] - s[:, 0])))
print("p90    cycles:", ", ".join(f"{n} {int(np.percentile(d[:, i], 90))}" for i, n in enumerate(names)))
st_rel = np.sort(s[:, 0] - t0)
print("block start times (cycles) deciles:", [int(v) for v in np.percentile(st_rel, [0, 10, 25, 50, 75, 90, 100])])
print("sum of block lifetimes / (span * 512 slots): %.2f" % ((s[:, 5] - s[:, 0]).sum() / (span * 512.0)))
