"""sf_qkv_time_attention2 (temporal qkv + time attention on the 192 x 384 main loop, + side gather / side GEMM / combine) against sf_qkv_time_attention (round 3, + CLS
GEMM / combine) at the model's size (run on the GPU box):   python tools/bench_qkv_time2.py [n_segments]"""
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
    out = torch.empty(rows, D, device=dev, dtype=torch.bfloat16)
    part = torch.empty(n * 12 * 49 * 66, device=dev)
    side_in = torch.empty(n * 33, D, device=dev, dtype=torch.bfloat16)
    side = torch.empty(n * 33, 3 * D, device=dev, dtype=torch.bfloat16)
    qkv_cls = torch.empty(n, 3 * D, device=dev, dtype=torch.bfloat16)

    def old():
        ops.gemm(x.view(n, L, D)[:, 0], w, b, qkv_cls)
        ops.qkv_time_attention(x, w, b, qkv_cls, out, part, n_seq=n, n_groups=196, scale=0.125)
        ops.attention_cls_combine(part, out, n_part=49, n_seq=n, out_seq_rows=L, out_row=0, heads=12)

    def new():
        ops.space_side_rows(x, side_in, n)
        ops.gemm(side_in, w, b, side)
        ops.qkv_time_attention2(x, w, b, side, out, part, n_seq=n, scale=0.125)
        ops.attention_cls_combine(part, out, n_part=33, n_seq=n, out_seq_rows=L, out_row=0, heads=12)

    def old_kernel():
        ops.qkv_time_attention(x, w, b, qkv_cls, out, part, n_seq=n, n_groups=196, scale=0.125)

    def new_kernel():
        ops.qkv_time_attention2(x, w, b, side, out, part, n_seq=n, scale=0.125)
    old(); new()
    rounds = []
    for _ in range(5):                                     # interleaved rounds, median
        rounds.append((timeit(old), timeit(new), timeit(old_kernel), timeit(new_kernel)))
    med = [sorted(r[i] for r in rounds)[2] for i in range(4)]
    flop = 2.0 * rows * 2304 * 768
    print(f'n_seg {n}: round-3 sequence (CLS gemm + sf_qkv_time_attention + combine) {med[0]:.1f} us | round-4 sequence (side gather + side gemm + sf_qkv_time_attention2 + '
          f'combine) {med[1]:.1f} us | kernels alone: {med[2]:.1f} us ({flop / med[2] / 1e6:.0f} TF on the GEMM FLOPs) -> {med[3]:.1f} us ({flop / med[3] / 1e6:.0f} TF)')


if __name__ == '__main__':
    main()
