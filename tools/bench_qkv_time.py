"""Micro-benchmark of the fused temporal qkv + time-attention launch against the un-fused sequence (run on the GPU box): python tools/bench_qkv_time.py [n_segments]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import ops, _lib

dev = torch.device('cuda:0')
L, D = 1569, 768


def timeit(fn, iters=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 224
    rows = n * L
    x = torch.randn(rows, D, device=dev).bfloat16()
    w, b = (torch.randn(3 * D, D, device=dev) * 0.02).bfloat16(), torch.randn(3 * D, device=dev) * 0.1
    qkv = torch.empty(rows, 3 * D, device=dev, dtype=torch.bfloat16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    out = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
    qkv_cls = torch.empty(n, 3 * D, device=dev, dtype=torch.bfloat16)
    part = torch.empty(n * 12 * 196 * 66, device=dev)

    def unfused():
        ops.gemm(x, w, b, qkv)
        ops.attention(q, k, v, out, n_seq=n, seq_rows=L, cls_row=0, heads=12, head_dim=64, scale=0.125, n_groups=196, row0=1, group_stride=1, tok_stride=196, n_tok=8)
        ops.attention_cls(q, k, v, out, n_seq=n, q_seq_rows=L, q_row=0, kv_seq_rows=L, kv_row0=0, n_keys=L, out_seq_rows=L, out_row=0, heads=12, head_dim=64, scale=0.125)

    def fused():
        ops.gemm(x.view(n, L, D)[:, 0], w, b, qkv_cls)
        ops.qkv_time_attention(x, w, b, qkv_cls, out, part, n_seq=n, n_groups=196, scale=0.125)
        ops.attention_cls_combine(part, out, n_part=49, n_seq=n, out_seq_rows=L, out_row=0, heads=12)

    t = {'gemm': [], 'unfused': [], 'fused': [], 'fused_kernel': [], 'r2_kernel': []}
    lib = _lib.load()
    for _ in range(5):
        t['gemm'].append(timeit(lambda: ops.gemm(x, w, b, qkv)))
        t['unfused'].append(timeit(unfused))
        t['fused'].append(timeit(fused))
        t['fused_kernel'].append(timeit(lambda: ops.qkv_time_attention(x, w, b, qkv_cls, out, part, n_seq=n, n_groups=196, scale=0.125)))
        lib.sf_qkv_time_force_schedule(0)
        t['r2_kernel'].append(timeit(lambda: ops.qkv_time_attention(x, w, b, qkv_cls, out, part, n_seq=n, n_groups=196, scale=0.125)))
        lib.sf_qkv_time_force_schedule(-1)
    med = {k_: sorted(v_)[len(v_) // 2] for k_, v_ in t.items()}
    fl = 2.0 * rows * 3 * D * D
    print(f"n_seg {n}: qkv GEMM alone {med['gemm']:7.1f} us ({fl / med['gemm'] / 1e6:4.0f} TF) | un-fused qkv + time attention + CLS {med['unfused']:7.1f} us | "
          f"fused (CLS-row GEMM + kernel + combine) {med['fused']:7.1f} us, kernel alone {med['fused_kernel']:7.1f} us ({fl / med['fused_kernel'] / 1e6:4.0f} TF) | round-2 loop kernel {med['r2_kernel']:7.1f} us")


if __name__ == '__main__':
    main()
