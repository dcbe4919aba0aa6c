"""What would a row split buy the Stage-1 GEMMs?  M = 43,932 rows (2 clips x 14 segments x 1,569 tokens) is 171.6 row panels of 256: the persistent 256 x 256 kernel
gives seven XCDs 22 panels each - 66 / 198 / 264 tiles for N = 768 / 2304 / 3072 on 32 CUs = 2.06 / 6.2 / 8.25 rounds, i.e. a whole extra round for a handful of tiles.
Times one launch (automatic configuration) against config 11 on the first 8 x 21 panels (43,008 rows: 63 / 189 / 252 tiles per XCD = 2 / 6 / 8 rounds) + the 128 x 128
kernel on the last 924 rows.

    python tools/bench_tail_split.py [rows]
"""
import sys

import torch

sys.path.insert(0, '.')
from synchformer_amd import _lib, ops                                   # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 43932
    dev = torch.device('cuda:0')
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(1)
    main_rows = (M // 2048) * 2048
    print(f'M {M}: main {main_rows} rows + tail {M - main_rows}')
    for (N, K, res, f32) in [(768, 768, False, False), (768, 768, True, True), (768, 2304, False, False), (768, 3072, False, False), (768, 3072, True, True),
                             (2304, 768, False, False), (3072, 768, False, False)]:
        a = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
        w = (torch.randn(N, K, device=dev, generator=g) * 0.05).bfloat16()
        b = torch.randn(N, device=dev, generator=g)
        r = torch.randn(M, N, device=dev, generator=g) if res else None
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        out2 = torch.empty_like(out)
        cfg = lib.sf_gemm_bf16_auto_config(M, N, K, 1 if res else 0)

        def whole():
            ops.gemm(a, w, b, out, residual=r)

        def split():
            lib.sf_gemm_force_config(11)
            ops.gemm(a[:main_rows], w, b, out2[:main_rows], residual=r[:main_rows] if res else None)
            lib.sf_gemm_force_config(0)
            ops.gemm(a[main_rows:], w, b, out2[main_rows:], residual=r[main_rows:] if res else None)
            lib.sf_gemm_force_config(-1)

        def whole11():
            lib.sf_gemm_force_config(11)
            ops.gemm(a, w, b, out, residual=r)
            lib.sf_gemm_force_config(-1)

        t_w, t_11, t_s = timed(whole), timed(whole11), timed(split)
        whole(); split()
        d = (out.float() - out2.float()).abs().max().item()
        print(f'N {N:5d} K {K:5d} res {int(res)} out {"f32" if f32 else "bf16"}: auto (config {cfg}) {t_w:7.1f} us   config 11 {t_11:7.1f} us   split {t_s:7.1f} us   max |diff| {d:.3e}')


if __name__ == '__main__':
    main()
