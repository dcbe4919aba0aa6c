"""sf_attention_group_bwd alone at the Stage-1 shape (n segments x 8 frames x 12 heads units of 196 queries x 197 keys): python tools/bench_group_bwd.py [n_segments]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = torch.device('cuda:0')
L, Dm = 1569, 768
M = n * L
g = torch.Generator(device=dev).manual_seed(3)
qkv = (torch.randn(M, 3 * Dm, device=dev, generator=g) * 0.5).bfloat16()
dO = (torch.randn(M, Dm, device=dev, generator=g) * 0.1).bfloat16()
d = torch.zeros(M, 3 * Dm, device=dev, dtype=torch.bfloat16)
cls_part = torch.zeros(n * 8, 2 * Dm, device=dev, dtype=torch.bfloat16)
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream


def run():
    rc = lib.sf_attention_group_bwd(qkv.data_ptr(), qkv[:, Dm:].data_ptr(), qkv[:, 2 * Dm:].data_ptr(), 3 * Dm, dO.data_ptr(), Dm, d.data_ptr(), d[:, Dm:].data_ptr(),
                                    d[:, 2 * Dm:].data_ptr(), 3 * Dm, cls_part.data_ptr(), n, L, 8, 1, 196, 1, 196, 0, 12, 64, 0.125, st)
    assert rc == 0, lib.sf_last_error()


for _ in range(3):
    run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 100)
ts.sort()
units = n * 8 * 12
flops = units * 5 * 2 * 197 * 196 * 64
print(f'group backward, {n} segments ({units} units): {ts[2]:.1f} us  ({flops / ts[2] / 1e6:.0f} TFLOP/s on the 5 necessary products)')
