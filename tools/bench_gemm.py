"""GEMM micro-benchmark on the model's real shapes (run on the GPU box): both tile configs, HIP-event timed.
    python tools/bench_gemm.py [n_segments ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import ops, _lib

dev = torch.device('cuda:0')
import os
ITERS = int(os.environ.get('ITERS', '6'))
CFGS = tuple(int(c) for c in os.environ.get('CFGS', '7,11').split(','))
lib = _lib.load_ablation()          # the ablation build: carries every tile config (the product library holds 0, 4, 7, 11 only)
_lib.using(lib).__enter__()


def timeit(fn, iters=20):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    segs = [int(a) for a in sys.argv[1:]] or [28]
    for n in segs:
        M = n * 1569
        print(f'--- n_seg {n}  M {M}')
        only = os.environ.get('SHAPES')
        for name, N, K, out_dt, gelu, res in [('qkv', 2304, 768, torch.bfloat16, False, False),
                                              ('proj+res', 768, 768, torch.float32, False, True),
                                              ('fc1+gelu', 3072, 768, torch.bfloat16, True, False),
                                              ('fc2+res', 768, 3072, torch.float32, False, True),
                                              # the data-gradient products of the Stage-1 backward (named in SHAPES only)
                                              ('dproj', 768, 768, torch.bfloat16, False, False),
                                              ('dqkv', 768, 2304, torch.float32, False, False),
                                              ('dfc1', 768, 3072, torch.float32, False, False),
                                              ('dfc2', 3072, 768, torch.bfloat16, False, False)]:
            if (only and name not in only.split(',')) or (not only and name.startswith('d')):
                continue
            a = torch.randn(M, K, device=dev).bfloat16()
            w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
            if os.environ.get('DATA') == 'zero':       # all-zero operands: the same instruction stream at a fraction of the switching power (is the kernel power-limited?)
                a.zero_(); w.zero_()
            b = torch.randn(N, device=dev)
            out = torch.zeros(M, N, device=dev, dtype=out_dt)
            line = f'{name:9s} N {N:4d} K {K:4d}: '
            # interleaved rounds (guide rule 24): every config is timed in every round, report the median
            times = {cfg: [] for cfg in CFGS}
            for _ in range(int(os.environ.get('ROUNDS', '7'))):
                for cfg in CFGS:
                    lib.sf_gemm_force_config(cfg)
                    times[cfg].append(timeit(lambda: ops.gemm(a, w, b, out, gelu=gelu, residual=out if res else None), iters=ITERS))
            for cfg in CFGS:
                us = sorted(times[cfg])[len(times[cfg]) // 2]
                line += f' c{cfg} {us:6.1f}us {2.0 * M * N * K / us / 1e6:5.0f}TF |'
            if os.environ.get('KMAJOR', '1') != '0' and M >= 8192:
                lib.sf_gemm_force_config(-1)
                wk = ops.ktile_major_weight(w)
                tk = sorted(timeit(lambda: ops.gemm(a, wk, b, out, gelu=gelu, residual=out if res else None), iters=6) for _ in range(7))[3]
                line += f' c7 k-tile-major W {tk:6.1f}us {2.0 * M * N * K / tk / 1e6:5.0f}TF |'
            print(line, flush=True)
    lib.sf_gemm_force_config(-1)


if __name__ == '__main__':
    main()
