"""Host logic of the row-walker weight gradient (csrc/wgrad_rows_kernel.hip, conv_api.hip) -- no device calls: which descriptors
it takes, and how the pixel splits of several problems that share launches shrink the fp32 slab workspace."""
import ctypes as C

from masterthesis_amd import _lib as L


def _desc(N, Ci, H, W, Co, k=3, stride=1, pad=1, transposed=0, dtype=L.MT_BF16, pad_mode=L.PAD_REFLECT, out_pad=0):
    return L.ConvDesc(dtype, transposed, N, H, W, Ci, Co, k, k, stride, pad, pad_mode, out_pad, L.ACT_NONE, 0.0)


def test_which_descriptors_the_row_walker_takes():
    lib = L.load()
    ok = lambda d: bool(lib.mt_conv_bwd_weight_rows_ok(C.byref(d)))
    assert ok(_desc(16, 64, 128, 128, 64))
    assert ok(_desc(16, 64, 256, 256, 128, stride=2))
    assert ok(_desc(16, 128, 128, 128, 64, stride=2, transposed=1, pad_mode=L.PAD_ZERO, out_pad=1))
    assert ok(_desc(16, 256, 32, 32, 256))                                   # a ping-pong shape on a small map
    assert not ok(_desc(16, 256, 64, 64, 256))                               # K1: the 256 x 256 ping-pong kernel
    assert not ok(_desc(16, 64, 128, 128, 128, k=4, stride=2))               # 4x4 window
    assert not ok(_desc(16, 48, 128, 128, 64))                               # channel count off the 64-wide blocks
    assert not ok(_desc(16, 64, 128, 48, 64))                                # not whole 32-pixel strips
    assert not ok(_desc(16, 64, 128, 128, 64, dtype=L.MT_F32))               # bf16 storage only
    assert not ok(_desc(16, 64, 129, 128, 128, stride=2))                    # odd height at stride 2: Hi != 2 Ho


def test_shared_launches_need_fewer_slabs_than_launches_of_their_own():
    lib = L.load()
    layers = [_desc(16, 64, 256, 256, 128, stride=2), _desc(16, 128, 128, 128, 256, stride=2), _desc(16, 64, 128, 128, 64),
              _desc(16, 64, 128, 128, 128), _desc(16, 128, 64, 64, 128), _desc(16, 128, 64, 64, 256),
              _desc(16, 256, 64, 64, 128, stride=2, transposed=1, pad_mode=L.PAD_ZERO, out_pad=1),
              _desc(16, 128, 128, 128, 64, stride=2, transposed=1, pad_mode=L.PAD_ZERO, out_pad=1)]
    alone = []
    for d in layers:
        one = (L.ConvDesc * 1)(d)
        b = int(lib.mt_conv_bwd_weight_rows_multi_ws_bytes(1, one))
        slab = 4 * 9 * d.Ci * d.Co
        assert b > 0 and b % slab == 0
        blocks = (d.Ci // 64) * (d.Co // 64)
        assert 1 <= (b // slab) * blocks <= 256, "a problem alone: at most one workgroup per compute unit"
        alone.append(b)
    arr = (L.ConvDesc * len(layers))(*layers)
    shared = int(lib.mt_conv_bwd_weight_rows_multi_ws_bytes(len(layers), arr))
    assert 0 < shared < sum(alone) // 3, (shared, sum(alone))
    # every problem keeps at least one slab
    assert shared >= sum(4 * 9 * d.Ci * d.Co for d in layers)
    # a descriptor the kernel does not take makes the whole call fail (0 bytes + error text), it is never silently dropped
    bad = (L.ConvDesc * 2)(layers[0], _desc(16, 64, 128, 128, 128, k=4, stride=2))
    assert int(lib.mt_conv_bwd_weight_rows_multi_ws_bytes(2, bad)) == 0
    assert b"row-walker" in lib.mt_last_error()
