"""Reference point only (never on the product path): what the vendor GEMM library behind torch.mm reaches on the model's shapes,
to put the hand-written sf_gemm_bf16 numbers in context.   python tools/bench_library_gemm.py [n_seg]"""
import sys
import torch

dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 224
M = n * 1569
for name, N, K in [('qkv', 2304, 768), ('proj', 768, 768), ('fc1', 3072, 768), ('fc2', 768, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        torch.mm(a, w.t(), out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        torch.mm(a, w.t(), out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f'{name:5s} M {M} N {N:4d} K {K:4d}: torch.mm (bf16 out, no bias/epilogue) {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TF')

# the same products through sf_gemm_bf16 (bf16 out, no bias / residual / GELU): apples to apples with the library line above
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from synchformer_amd import ops  # noqa: E402
for name, N, K in [('qkv', 2304, 768), ('proj', 768, 768), ('fc1', 3072, 768), ('fc2', 768, 3072)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, None, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        ops.gemm(a, w, None, out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f'{name:5s} M {M} N {N:4d} K {K:4d}: sf_gemm_bf16 (bf16 out, no epilogue extras)      {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.0f} TF')
