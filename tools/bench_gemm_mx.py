"""MXFP8 GEMM micro-benchmark on the model's shapes against the bf16 kernel (run on the GPU box): python tools/bench_gemm_mx.py [n_segments ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import ops, _lib
lib = _lib.load()

dev = torch.device('cuda:0')


def timeit(fn, iters=6):
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
    for n in [int(a) for a in sys.argv[1:]] or [208]:
        M = n * 1569
        print(f'--- n_seg {n}  M {M}')
        for name, N, K, out_dt, gelu, res in [('qkv', 2304, 768, torch.bfloat16, False, False), ('proj+res', 768, 768, torch.float32, False, True),
                                              ('fc1+gelu', 3072, 768, torch.bfloat16, True, False), ('fc2+res', 768, 3072, torch.float32, False, True)]:
            a = torch.randn(M, K, device=dev).bfloat16()
            w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
            b = torch.randn(N, device=dev)
            out = torch.zeros(M, N, device=dev, dtype=out_dt)
            aq, asc = torch.empty(M, K, device=dev, dtype=torch.uint8), ops.mx_scale_planes(M, K, dev)
            wq, wsc = torch.empty(N, K, device=dev, dtype=torch.uint8), ops.mx_scale_planes(N, K, dev)
            ops.quantize_mxfp8(w, wq, wsc)
            oq = torch.empty(M, N, device=dev, dtype=torch.uint8) if (not res and N % 128 == 0) else None       # MXFP8 output (the next GEMM's A operand)
            osc = ops.mx_scale_planes(M, N, dev) if oq is not None else None
            t = {'bf16': [], 'quant': [], 'mx': [], 'mxq': [], 'mx_r2': []}
            for _ in range(5):
                t['bf16'].append(timeit(lambda: ops.gemm(a, w, b, out, gelu=gelu, residual=out if res else None)))
                t['quant'].append(timeit(lambda: ops.quantize_mxfp8(a, aq, asc)))
                t['mx'].append(timeit(lambda: ops.gemm_mxfp8(aq, asc, wq, wsc, b, out, gelu=gelu, residual=out if res else None)))
                lib.sf_gemm_mx_force_schedule(0)
                t['mx_r2'].append(timeit(lambda: ops.gemm_mxfp8(aq, asc, wq, wsc, b, out, gelu=gelu, residual=out if res else None)))
                lib.sf_gemm_mx_force_schedule(-1)
                t['mxq'].append(timeit(lambda: ops.gemm_mxfp8(aq, asc, wq, wsc, b, oq, gelu=gelu, out_scales=osc)) if oq is not None else 0.0)
            med = {k: sorted(v)[len(v) // 2] for k, v in t.items()}
            fl = 2.0 * M * N * K
            extra = ''
            if res:                                                   # the fused proj / fc2 + residual + LayerNorm + quantisation against the pair it replaces
                gam, bet = torch.ones(768, device=dev), torch.zeros(768, device=dev)
                yq, ysc = torch.empty(M, 768, device=dev, dtype=torch.uint8), ops.mx_scale_planes(M, 768, dev)
                tf, tl = [], []
                for _ in range(5):
                    tf.append(timeit(lambda: ops.gemm_mx_res_ln(aq, asc, wq, wsc, b, out, gam, bet, yq, ysc, 1e-6)))
                    tl.append(timeit(lambda: ops.layernorm_mxfp8(out, gam, bet, yq, ysc, 1e-6)))
                tf, tl = sorted(tf)[2], sorted(tl)[2]
                extra = f' | + LayerNorm -> MXFP8: separate {tl:6.1f} us, fused sf_gemm_mx_res_ln768 {tf:7.1f} us ({fl / tf / 1e6:5.0f} TF) vs pair {med["mx"] + tl:7.1f} us'
            print(f"{name:9s} N {N:4d} K {K:4d}: bf16 {med['bf16']:7.1f} us ({fl / med['bf16'] / 1e6:5.0f} TF) | mxfp8 {med['mx']:7.1f} us ({fl / med['mx'] / 1e6:5.0f} TF)"
                  f" + quantise A {med['quant']:6.1f} us | MXFP8 out {med['mxq']:7.1f} us | round-2 loop {med['mx_r2']:7.1f} us ({fl / med['mx_r2'] / 1e6:5.0f} TF)" + extra, flush=True)


if __name__ == '__main__':
    main()
