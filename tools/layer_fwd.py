"""20 forward launches of one layer (for rocprofv3 passes): layer_fwd.py N Cin H Cout k stride [T]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from masterthesis_amd import hip_ops as ops
dev = torch.device('cuda:0')
ops.set_compute_dtype(torch.bfloat16)
N, Ci, H, Co, k, st = [int(a) for a in sys.argv[1:7]]
tr = len(sys.argv) > 7 and sys.argv[7] == "T"
x = ops.canon(torch.randn(N, Ci, H, H, device=dev))
w = torch.randn(*((Ci, Co, k, k) if tr else (Co, Ci, k, k)), device=dev) * 0.05
with torch.no_grad():
    for _ in range(20):
        y = ops.conv_transpose2d(x, w, None, stride=st, pad=1, out_pad=1) if tr else \
            ops.conv2d(x, w, None, stride=st, pad=k // 2, pad_mode="reflect")
torch.cuda.synchronize()
