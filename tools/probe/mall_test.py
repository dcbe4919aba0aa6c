import torch
dev = torch.device('cuda:0')
def t(fn, n=5):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
N = 540 * 1024 * 1024 // 2            # 0.54 GB of bf16
a = torch.empty(N, device=dev, dtype=torch.bfloat16)
other = torch.empty(N * 2, device=dev, dtype=torch.bfloat16)
chunk = 100 * 1024 * 1024 // 2        # 100 MB
out = torch.empty(chunk, device=dev, dtype=torch.bfloat16)
for name, sl in (('oldest 100 MB', slice(0, chunk)), ('newest 100 MB', slice(N - chunk, N)), ('middle', slice(N // 2, N // 2 + chunk))):
    def run():
        a.fill_(1.0)                  # streaming write of the whole buffer (the producer kernel)
        torch.cuda.synchronize()
    def read():
        torch.add(a[sl], 1.0, out=out)
    res = []
    for _ in range(5):
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); read(); e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1) * 1e3)
    print(f'{name}: read+write of 100 MB right after writing 540 MB: {sorted(res)[2]:.1f} us  ({2 * 100 / sorted(res)[2] * 1e6 / 1e6 / 1e3 * 1e3:.0f} GB/s eff)')
# control: after flushing with another big buffer
res = []
for _ in range(5):
    a.fill_(1.0); other.fill_(2.0); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.add(a[N - chunk:N], 1.0, out=out); e1.record(); torch.cuda.synchronize(); res.append(e0.elapsed_time(e1) * 1e3)
print(f'newest 100 MB after 1.08 GB of other writes: {sorted(res)[2]:.1f} us')
