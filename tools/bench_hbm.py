"""What the device sustains for pure streaming writes / copies / reads (torch kernels): the ceiling for the
write-bound epilogues and the elementwise family."""
import torch
dev = torch.device('cuda:0')
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3
for mb in (32, 128, 512, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev, dtype=torch.float32)
    y = torch.empty(n, device=dev, dtype=torch.float32)
    tw = t(lambda: x.zero_())
    tc = t(lambda: y.copy_(x))
    tr = t(lambda: x.sum())
    print(f"{mb:5d} MB: write {mb/1024/1024/tw*1024*1024/1e6:.2f} TB/s  copy (r+w) {2*mb/1e6/tc:.2f} TB/s  read {mb/1e6/tr:.2f} TB/s")
