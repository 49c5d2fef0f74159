"""K1 (3x3 s1 256->256 @64x64, reflect) weight gradients through mt_conv_bwd_weight_group (G problems per launch) and one by one,
for rocprofv3 --kernel-trace --stats:  python tools/bench_wgrad_group.py [N] [G]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import _lib as L
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
lib = L.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4
desc = L.ConvDesc(L.MT_BF16, 0, N, 64, 64, 256, 256, 3, 3, 1, 1, L.PAD_REFLECT, 0, L.ACT_NONE, 0.0)
xs = [ops.canon(torch.randn(N, 256, 64, 64, device=dev)) for _ in range(G)]
dys = [ops.canon(torch.randn(N, 256, 64, 64, device=dev)) for _ in range(G)]
dws = [torch.zeros(256, 256, 3, 3, device=dev) for _ in range(G)]
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
nws = int(lib.mt_conv_bwd_weight_ws_bytes(C.byref(desc)))
ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
nwg = int(lib.mt_conv_bwd_weight_group_ws_bytes(C.byref(desc), G))
wsg = torch.empty((max(nwg, 16),), dtype=torch.uint8, device=dev)
xa = (C.c_void_p * G)(*[t.data_ptr() for t in xs])
da = (C.c_void_p * G)(*[t.data_ptr() for t in dys])
wa = (C.c_void_p * G)(*[t.data_ptr() for t in dws])


def single():
    for g in range(G):
        L.check(lib.mt_conv_bwd_weight(C.byref(desc), P(xs[g]), P(dys[g]), P(dws[g]), None, P(ws), nws, 1, st), "single")


def group():
    L.check(lib.mt_conv_bwd_weight_group(C.byref(desc), G, xa, da, wa, P(wsg), nwg, 1, st), "group")


for name, fn in (("single", single), ("group", group)):
    if name == "group" and nwg == 0:
        print("no group launch for this shape"); break
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 10 / G * 1e3:.1f} us per problem (N={N}, G={G})", flush=True)
