"""Micro-benchmark of the fused full-row GEMM + residual + LayerNorm kernel against the un-fused pair (run on the GPU box).
    python tools/bench_gemm_ln.py [n_segments ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import ops, _lib
import os

dev = torch.device('cuda:0')
_lib.using(_lib.load_ablation()).__enter__()      # schedule 2 lives in the ablation build (the product library refuses it)


def timeit(fn, iters=6):
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
    segs = [int(a) for a in sys.argv[1:]] or [224]
    for n in segs:
        M = n * 1569
        print(f'--- n_seg {n}  M {M}')
        for name, K in (('proj', 768), ('fc2', 3072)):
            a = torch.randn(M, K, device=dev).bfloat16()
            w = (torch.randn(768, K, device=dev) * 0.02).bfloat16()
            b, g, bt = torch.randn(768, device=dev), torch.randn(768, device=dev), torch.randn(768, device=dev)
            x = torch.randn(M, 768, device=dev)
            y = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
            wk = ops.kmajor_weight(w)
            t = {'gemm': [], 'ln': [], 'fused': [], 'fused_k': [], 'r2_k': [], 's2': []}
            lib = _lib.load()
            for _ in range(7):      # interleaved rounds, median
                t['gemm'].append(timeit(lambda: ops.gemm(a, w, b, x, residual=x)))
                t['ln'].append(timeit(lambda: ops.layernorm(x, g, bt, y, 1e-6)))
                t['fused'].append(timeit(lambda: ops.gemm_res_ln(a, w, b, x, g, bt, y, 1e-6)))
                t['fused_k'].append(timeit(lambda: ops.gemm_res_ln(a, wk, b, x, g, bt, y, 1e-6)))
                lib.sf_gemm_res_ln_force_schedule(0)
                try:                                                   # (ablation builds exist for one schedule or the other)
                    t['r2_k'].append(timeit(lambda: ops.gemm_res_ln(a, wk, b, x, g, bt, y, 1e-6)))
                except RuntimeError:
                    t['r2_k'].append(float('nan'))
                lib.sf_gemm_res_ln_force_schedule(2)
                t['s2'].append(timeit(lambda: ops.gemm_res_ln(a, w, b, x, g, bt, y, 1e-6)))
                lib.sf_gemm_res_ln_force_schedule(-1)
                x.normal_()
            med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
            fl = 2.0 * M * 768 * K
            byts = M * K * 2 + M * 768 * (4 + 4 + 2)
            print(f"{name:5s} K {K:4d}: gemm+res {med['gemm']:7.1f} us ({fl / med['gemm'] / 1e6:5.0f} TF) + layernorm {med['ln']:6.1f} us = "
                  f"{med['gemm'] + med['ln']:7.1f} us | fused {med['fused']:7.1f} us ({fl / med['fused'] / 1e6:5.0f} TF, {byts / med['fused'] / 1e6:5.2f} TB/s algorithmic) | k-major W {med['fused_k']:7.1f} us ({fl / med['fused_k'] / 1e6:5.0f} TF) | round-2 loop, k-major W {med['r2_k']:7.1f} us | schedule 2 (192-row tiles, two column passes) {med['s2']:7.1f} us ({fl / med['s2'] / 1e6:5.0f} TF)",
                  flush=True)


if __name__ == '__main__':
    main()
