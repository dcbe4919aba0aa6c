"""Weight-gradient product micro-benchmark (run on the GPU box): sf_gemm_tn_splitk on the row-major operands vs the transposed-copy path.
    python tools/bench_wgrad.py [rows]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import _lib
from synchformer_amd import train as T

dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 43932


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


st = torch.cuda.current_stream().cuda_stream
for N, K in [(768, 768), (2304, 768), (3072, 768), (768, 3072)]:
    dy = torch.randn(M, N, device=dev).bfloat16()
    x = torch.randn(M, K, device=dev).bfloat16()
    tiles = (N // 128) * (K // 128)
    flop = 2.0 * M * N * K
    for split in sorted({T._wgrad_split(tiles), max(1, min(32, 768 // tiles))}):   # the trainer's fill-aware choice vs the first rule
        m_pad = ((M + 63) // 64) * 64
        kc = ((m_pad // split + 63) // 64) * 64
        part = torch.empty(split * N, K, device=dev)
        t_tn = timeit(lambda: _lib.check(_lib.load().sf_gemm_tn_splitk(dy.data_ptr(), N, x.data_ptr(), K, part.data_ptr(), None, M, N, K, split, kc, st), 'tn'))
        mp = kc * split
        dyT = torch.zeros(N, mp, device=dev, dtype=torch.bfloat16)
        xT = torch.zeros(K, mp, device=dev, dtype=torch.bfloat16)
        t_tr = timeit(lambda: (T.transpose(dy, N, 0, 0, dyT, mp, 0, 0, M, N, mp), T.transpose(x, K, 0, 0, xT, mp, 0, 0, M, K, mp)))
        t_nn = timeit(lambda: T.bgemm(dyT, mp, kc, 0, xT, mp, kc, 0, part, K, N * K, 0, N, K, kc, split, 1))
        # the quadrant-phased 256 x 256 kernel: as many chunks as fill the chip once with 256 x 256 tiles
        t256 = (N // 256) * (K // 256)
        sp = max(1, torch.cuda.get_device_properties(0).multi_processor_count // t256)
        kc2 = ((M + sp - 1) // sp + 127) // 128 * 128
        sp = (M + kc2 - 1) // kc2
        part2 = torch.empty(sp * N, K, device=dev)
        bp2 = torch.empty(sp, N, device=dev)
        t_pp = timeit(lambda: _lib.check(_lib.load().sf_gemm_tn_pp(dy.data_ptr(), N, x.data_ptr(), K, part2.data_ptr(), bp2.data_ptr(), M, N, K, sp, kc2, st), 'tn_pp'))
        t_pp0 = timeit(lambda: _lib.check(_lib.load().sf_gemm_tn_pp(dy.data_ptr(), N, x.data_ptr(), K, part2.data_ptr(), None, M, N, K, sp, kc2, st), 'tn_pp'))
        outw = torch.empty(N, K, device=dev)
        t_s1 = timeit(lambda: _lib.load().sf_seqsum(part.data_ptr(), K, split, N, K, outw.data_ptr(), 0, st))
        t_s2 = timeit(lambda: _lib.load().sf_seqsum(part2.data_ptr(), K, sp, N, K, outw.data_ptr(), 0, st))
        print(f'    quadrant-phased 256x256, split {sp:3d}: {t_pp:7.1f} us ({flop / t_pp / 1e6:6.0f} TF), without the bias sums {t_pp0:7.1f} us | chunk sums: {t_s1:5.1f} us (split {split}) vs {t_s2:5.1f} us (split {sp})')
        print(f'N={N:5d} K={K:5d} split={split:3d}: TN {t_tn:7.1f} us ({flop / t_tn / 1e6:6.0f} TF)   NN {t_nn:7.1f} us ({flop / t_nn / 1e6:6.0f} TF) + transposes {t_tr:6.1f} us')
