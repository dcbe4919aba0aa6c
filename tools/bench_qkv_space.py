"""Fused spatial qkv + space attention (sf_qkv_space_attention + its side GEMM + combine) against the un-fused launches it replaces, at the model's size
(run on the GPU box):   python tools/bench_qkv_space.py [n_segments]"""
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 224
    L, D = 1569, 768
    rows = n * L
    x = torch.randn(rows, D, device=dev).bfloat16()
    w = (torch.randn(3 * D, D, device=dev) * 0.05).bfloat16()
    b = torch.randn(3 * D, device=dev) * 0.1
    qkv = torch.empty(rows, 3 * D, device=dev, dtype=torch.bfloat16)
    out = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
    part = torch.empty(n * 12 * 8 * 66, device=dev)
    side_in = torch.empty(n * 33, D, device=dev, dtype=torch.bfloat16)
    side = torch.empty(n * 33, 3 * D, device=dev, dtype=torch.bfloat16)
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]

    def unfused():
        ops.gemm(x, w, b, qkv)
        ops.attention_cls_partial(q, k, v, out, part, n_seq=n, seq_rows=L, n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196, cls_row=0, heads=12,
                                  head_dim=64, scale=0.125)
        ops.attention_cls_combine(part, out, n_part=8, n_seq=n, out_seq_rows=L, out_row=0, heads=12)

    def fused():
        ops.space_side_rows(x, side_in, n)
        ops.gemm(side_in, w, b, side)
        ops.qkv_space_attention(x, w, b, side, out, part, n_seq=n, scale=0.125)
        ops.attention_cls_combine(part, out, n_part=8, n_seq=n, out_seq_rows=L, out_row=0, heads=12)

    def kernel_only():
        ops.qkv_space_attention(x, w, b, side, out, part, n_seq=n, scale=0.125)
    fused()
    rounds = []
    for _ in range(5):                                     # interleaved rounds, median
        rounds.append((timeit(unfused), timeit(fused), timeit(kernel_only), timeit(lambda: ops.gemm(x, w, b, qkv))))
    med = [sorted(r[i] for r in rounds)[2] for i in range(4)]
    flop = 2.0 * rows * 2304 * 768
    print(f'n_seg {n}: un-fused (gemm + attention + combine) {med[0]:.1f} us | fused (side gather + side gemm + kernel + combine) {med[1]:.1f} us | '
          f'fused kernel alone {med[2]:.1f} us = {flop / med[2] / 1e6:.0f} TF on the GEMM FLOPs | qkv gemm alone {med[3]:.1f} us')


if __name__ == '__main__':
    main()
