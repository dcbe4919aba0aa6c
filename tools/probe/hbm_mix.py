"""What does a pure streaming kernel reach on the traffic mix of sf_gemm_res_ln768's proj form (3.24 GB: half read, half written)?  (run on the GPU box)"""
import torch
dev = torch.device('cuda:0')
n = 1620 * 1000 * 1000 // 4
x = torch.randn(n, device=dev)
y = torch.empty_like(x)
def t(fn, it=10):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / it * 1e3
us = t(lambda: torch.add(x, 1.0, out=y))
print(f'read 1.62 GB + write 1.62 GB (torch add, out of place): {us:.1f} us = {3.24e9 / us / 1e6:.2f} TB/s')
us = t(lambda: x.add_(1.0))
print(f'in place (read 1.62 + write 1.62 GB, same lines): {us:.1f} us = {3.24e9 / us / 1e6:.2f} TB/s')
us = t(lambda: y.copy_(x))
print(f'copy: {us:.1f} us = {3.24e9 / us / 1e6:.2f} TB/s')
s = torch.empty(1, device=dev)
us = t(lambda: torch.sum(x, dim=0, keepdim=True, out=s))
print(f'read only 1.62 GB: {us:.1f} us = {1.62e9 / us / 1e6:.2f} TB/s')
