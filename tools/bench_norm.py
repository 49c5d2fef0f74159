"""InstanceNorm+ReLU forward/backward on the K1 activation shape, for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import _lib
if os.environ.get('MT_DIAG_LIB'):
    _lib.LIB_PATH = os.path.abspath(os.environ['MT_DIAG_LIB'])
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
for N in (16, 32):
    x = ops.canon(torch.randn(N, 256, 64, 64, device=dev)).detach().requires_grad_()
    gy = ops.canon(torch.randn(N, 256, 64, 64, device=dev)).detach()
    for _ in range(10):
        x.grad = None
        y = ops.instance_norm_act(x, act="relu")
        y.backward(gy)
torch.cuda.synchronize()
