"""Times mt_conv_bwd_data and mt_conv_bwd_weight separately through the C ABI (HIP events)."""
import ctypes as C, sys
sys.path.insert(0, '.')
import torch
from masterthesis_amd import _lib as L, hip_ops as ops
lib = L.load(); dev = torch.device('cuda:0')
def run(name, N, Ci, H, W, Co, k, stride, pad, mode, iters=30):
    d = L.ConvDesc(L.MT_BF16, 0, N, H, W, Ci, Co, k, k, stride, pad, mode, 0, 0, 0.01)
    ho, wo = C.c_int(), C.c_int(); lib.mt_conv_out_hw(C.byref(d), C.byref(ho), C.byref(wo))
    Ho, Wo = ho.value, wo.value
    x = torch.randn(N, H, W, ops.padc(Ci), device=dev).bfloat16()
    dy = torch.randn(N, Ho, Wo, ops.padc(Co), device=dev).bfloat16()
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pk = torch.empty(lib.mt_conv_pack_bytes(C.byref(d), 1), dtype=torch.uint8, device=dev)
    lib.mt_conv_pack(C.byref(d), 1, C.c_void_p(w.data_ptr()), C.c_void_p(pk.data_ptr()), s)
    dx = torch.empty_like(x); dw = torch.empty_like(w)
    n1 = lib.mt_conv_bwd_data_ws_bytes(C.byref(d)); ws1 = torch.empty(max(n1, 16), dtype=torch.uint8, device=dev)
    n2 = lib.mt_conv_bwd_weight_ws_bytes(C.byref(d)); ws2 = torch.empty(n2, dtype=torch.uint8, device=dev)
    def fd(): lib.mt_conv_bwd_data(C.byref(d), C.c_void_p(dy.data_ptr()), C.c_void_p(pk.data_ptr()), C.c_void_p(dx.data_ptr()), C.c_void_p(ws1.data_ptr()), n1, s)
    def fw(): lib.mt_conv_bwd_weight(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(dy.data_ptr()), C.c_void_p(dw.data_ptr()), None, C.c_void_p(ws2.data_ptr()), n2, 0, s)
    def t(fn):
        for _ in range(3): fn()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters
    flop = 2.0 * N * Ho * Wo * Co * Ci * k * k
    td, tw = t(fd), t(fw)
    print(f"{name:28s} dgrad {td*1e3:7.1f} us {flop/td/1e9:6.0f} TF | wgrad {tw*1e3:7.1f} us {flop/tw/1e9:6.0f} TF")
run("K1 N16", 16, 256, 64, 64, 256, 3, 1, 1, 1)
run("K1 N8", 8, 256, 64, 64, 256, 3, 1, 1, 1)
run("down 128->256 s2 @128", 16, 128, 128, 128, 256, 3, 2, 1, 1)
run("down 64->128 s2 @256", 16, 64, 256, 256, 128, 3, 2, 1, 1)
run("stem 7x7 3->64", 16, 3, 256, 256, 64, 7, 1, 3, 1)
run("D 512->1024 s2 @16 N32", 32, 512, 16, 16, 1024, 3, 2, 1, 1)
run("D 64->128 s2 @128 N32", 32, 64, 128, 128, 128, 3, 2, 1, 1)
