"""Attention micro-benchmark on the model's shapes (run on the GPU box).  python tools/bench_attention.py [n_seg]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import ops

dev = torch.device('cuda:0')


def timeit(fn, iters=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


n = int(sys.argv[1]) if len(sys.argv) > 1 else 27
L = 1569
qkv = (torch.randn(n * L, 2304, device=dev) * 1.0).bfloat16()
out = torch.zeros(n * L, 768, device=dev, dtype=torch.bfloat16)
q, k, v = qkv[:, :768], qkv[:, 768:1536], qkv[:, 1536:]
mb = (qkv.numel() + out.numel()) * 2 / 1e6
t = timeit(lambda: ops.attention(q, k, v, out, n_seq=n, seq_rows=L, n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196,
                                 cls_row=0, heads=12, head_dim=64, scale=0.125))
print(f'space  (mfma) : {t:7.1f} us  {mb / t:6.2f} TB/s-equivalent of the {mb:.0f} MB algorithmic traffic')
t = timeit(lambda: ops.attention(q, k, v, out, n_seq=n, seq_rows=L, n_groups=196, row0=1, group_stride=1, tok_stride=196, n_tok=8,
                                 cls_row=0, heads=12, head_dim=64, scale=0.125))
print(f'time   (tiny) : {t:7.1f} us  {mb / t:6.2f} TB/s-equivalent')
t = timeit(lambda: ops.attention_cls(q, k, v, out, n_seq=n, q_seq_rows=L, q_row=0, kv_seq_rows=L, kv_row0=0, n_keys=L,
                                     out_seq_rows=L, out_row=0, heads=12, head_dim=64, scale=0.125))
print(f'cls row       : {t:7.1f} us  {qkv.numel() * 2 * 2 / 3 / 1e6 / t:6.2f} TB/s-equivalent (K+V once)')
na = n * 14 // 14
qa = (torch.randn(448 * 74, 2304, device=dev)).bfloat16()
oa = torch.zeros(448 * 74, 768, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: ops.attention(qa[:, :768], qa[:, 768:1536], qa[:, 1536:], oa, n_seq=448, seq_rows=74, n_groups=1, row0=0,
                                 group_stride=0, tok_stride=1, n_tok=74, cls_row=-1, heads=12, head_dim=64, scale=0.125))
print(f'AST 74 tokens x 448 seq: {t:7.1f} us')
# layout probe: the same space attention with every head's q | k | v rows contiguous ([seq x head][token][192], 384 B per token) instead of 128-byte
# slices of 4608-byte rows - emulated as 12 x n single-head sequences with ld = 192; and with q, k, v in three separate head-major planes (ld = 64)
hm = (torch.randn(n * 12 * L, 192, device=dev)).bfloat16()
ohm = torch.zeros(n * 12 * L, 64, device=dev, dtype=torch.bfloat16)
t = timeit(lambda: ops.attention(hm[:, :64], hm[:, 64:128], hm[:, 128:], ohm, n_seq=n * 12, seq_rows=L, n_groups=8, row0=1, group_stride=196, tok_stride=1,
                                 n_tok=196, cls_row=0, heads=1, head_dim=64, scale=0.125))
print(f'space, head-major q|k|v rows (ld 192): {t:7.1f} us  {mb / t:6.2f} TB/s-equivalent')
pl = (torch.randn(3, n * 12 * L, 64, device=dev)).bfloat16()
t = timeit(lambda: ops.attention(pl[0], pl[1], pl[2], ohm, n_seq=n * 12, seq_rows=L, n_groups=8, row0=1, group_stride=196, tok_stride=1,
                                 n_tok=196, cls_row=0, heads=1, head_dim=64, scale=0.125))
print(f'space, three head-major planes (ld 64): {t:7.1f} us  {mb / t:6.2f} TB/s-equivalent')
