"""sf_gemm_res_ln768 under one SF_RL_ABL mask (1 = no residual loads, 4 = no Y stores, 7 = none of the epilogue's memory traffic; ablation build.  Mask 2 alone - no X stores - is not meaningful: the counted vmcnt waits of the schedule assume them) - which part of the epilogue's memory traffic costs
what (run on the GPU box):   for a in 0 1 4 7; do SF_RL_ABL=$a python tools/r06_gemm_ln_abl.py; done"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import ops, _lib

dev = torch.device('cuda:0')
_lib.using(_lib.load_ablation()).__enter__()
M = 224 * 1569
out = []
for name, K in (('proj', 768), ('fc2', 3072)):
    a = torch.randn(M, K, device=dev).bfloat16()
    wk = ops.kmajor_weight((torch.randn(768, K, device=dev) * 0.02).bfloat16())
    b, g, bt = torch.randn(768, device=dev), torch.randn(768, device=dev), torch.randn(768, device=dev)
    x = torch.randn(M, 768, device=dev)
    y = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
    ts = []
    for _ in range(7):
        ops.gemm_res_ln(a, wk, b, x, g, bt, y, 1e-6)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(6):
            ops.gemm_res_ln(a, wk, b, x, g, bt, y, 1e-6)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 6 * 1e3)
        x.normal_()
    out.append(f'{name} {sorted(ts)[3]:7.1f} us')
print(f"SF_RL_ABL={os.environ.get('SF_RL_ABL', '0')}: " + ' | '.join(out))
