"""Streaming kernels of the normalisation family on the activation shapes of the step (bf16): HIP-event timings of
the statistics / apply passes forward and backward, with the achieved bytes/s next to them.
    python tools/bench_norm.py            (also usable under rocprofv3 --kernel-trace --stats)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import _lib as L
from masterthesis_amd import hip_ops as ops

dev = torch.device("cuda:0")
ops.set_compute_dtype(torch.bfloat16)
lib = L.load()


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def tbs(nb, us):
    return nb / us / 1e6


for (N, Cc, H, W, act, res) in [(16, 256, 64, 64, L.ACT_RELU, False), (16, 256, 64, 64, L.ACT_NONE, True),
                                (32, 256, 64, 64, L.ACT_RELU, False), (16, 128, 128, 128, L.ACT_RELU, False),
                                (16, 64, 256, 256, L.ACT_RELU, False), (16, 64, 128, 128, L.ACT_RELU, False)]:
    x = ops.canon(torch.randn(N, Cc, H, W, device=dev))
    dy = ops.canon(torch.randn(N, Cc, H, W, device=dev))
    r = ops.canon(torch.randn(N, Cc, H, W, device=dev)) if res else None
    y = torch.empty_like(x)
    Cp, HW = ops.padc(Cc), H * W
    nparts = int(lib.mt_nc_stats_parts(L.MT_BF16, N, HW, Cp))
    part = torch.empty((N, nparts, Cp, 2), dtype=torch.float32, device=dev)
    coef = torch.empty((4, N, Cp), dtype=torch.float32, device=dev)
    cc = torch.empty((3, N, Cp), dtype=torch.float32, device=dev)
    nbytes = x.numel() // Cc * Cp * 2
    t_s = timeit(lambda: lib.mt_nc_stats(L.MT_BF16, P(x), P(part), N, HW, Cp, st()))
    t_f = timeit(lambda: lib.mt_norm_finalize(L.NORM_INSTANCE, P(part), None, None, None, P(coef[0]), P(coef[1]),
                                              P(coef[2]), P(coef[3]), N, HW, Cc, Cp, 1e-5, nparts, st()))
    t_a = timeit(lambda: lib.mt_scale_shift_act(L.MT_BF16, P(x), P(coef[0]), P(coef[1]), P(r), P(y), N, HW, Cp, act,
                                                0.01, st()))
    t_sb = timeit(lambda: lib.mt_nc_stats_bwd(L.MT_BF16, P(dy), P(x), P(coef[0]), P(coef[1]), P(part), N, HW, Cp, act,
                                              0.01, st()))
    t_fb = timeit(lambda: lib.mt_norm_bwd_finalize(L.NORM_INSTANCE, P(part), P(coef[2]), P(coef[3]), None, None,
                                                   P(cc[0]), P(cc[1]), P(cc[2]), None, None, None, N, HW, Cc, Cp,
                                                   nparts, st()))
    t_ab = timeit(lambda: lib.mt_norm_bwd_apply(L.MT_BF16, P(dy), P(x), P(coef[0]), P(coef[1]), P(cc[0]), P(cc[1]),
                                                P(cc[2]), P(y), N, HW, Cp, act, 0.01, st()))
    sl = C.c_int(0)
    t_op = float("nan")
    if lib.mt_norm_bwd_onepass_ok(L.MT_BF16, L.NORM_INSTANCE, N, HW, Cp, act, 0, C.byref(sl)):
        part1 = torch.empty((N, sl.value, Cp, 2), dtype=torch.float32, device=dev)
        sync = torch.zeros(2 * N, dtype=torch.int32, device=dev)
        status = torch.zeros(4, dtype=torch.int32, device=dev)
        t_op = timeit(lambda: lib.mt_norm_bwd_onepass(L.MT_BF16, L.NORM_INSTANCE, P(dy), P(x), P(coef[0]), P(coef[1]), P(coef[2]),
                                                      P(coef[3]), None, None, None, None, None, P(y), P(part1), P(sync), P(status), 0, N, HW, Cc, Cp, act, 0.01,
                                                      st()))
        assert int(sync.abs().sum().item()) == 0 and int(status.abs().sum().item()) == 0
    print(f"   one-pass backward {t_op:.1f}us ({tbs(3 * nbytes, t_op):.2f} TB/s) vs three-pass {t_sb + t_fb + t_ab:.1f}us")
    print(f"[{N},{Cc},{H},{W}] res={int(res)} nparts={nparts}: stats {t_s:.1f}us ({tbs(nbytes, t_s):.2f} TB/s)  "
          f"finalize {t_f:.1f}us  apply {t_a:.1f}us ({tbs(nbytes * (3 if res else 2), t_a):.2f} TB/s) | bwd stats "
          f"{t_sb:.1f}us ({tbs(2 * nbytes, t_sb):.2f} TB/s)  finalize {t_fb:.1f}us  apply {t_ab:.1f}us "
          f"({tbs(3 * nbytes, t_ab):.2f} TB/s)", flush=True)
