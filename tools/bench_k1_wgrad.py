"""K1 (3x3 s1 256->256 @64x64, reflect) weight gradient alone (wgrad + unpack kernels), for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import _lib
if os.environ.get('MT_DIAG_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MT_DIAG_LIB'])
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
x = ops.canon(torch.randn(N, 256, 64, 64, device=dev)).detach()
w = (torch.randn(256, 256, 3, 3, device=dev) * 0.05).requires_grad_()
y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")
gy = ops.canon(torch.randn_like(y.float())).detach()
for _ in range(20):
    y = ops.conv2d(x, w, None, stride=1, pad=1, pad_mode="reflect")
    y.backward(gy)
torch.cuda.synchronize()
