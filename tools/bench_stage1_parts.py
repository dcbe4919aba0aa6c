"""Time the pieces of the Stage-1 backward on the GPU box: python tools/bench_stage1_parts.py [n_segments]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from synchformer_amd import synth
from synchformer_amd.stage1 import AVCLIPTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = torch.device('cuda:0')
sd = {k: v for k, v in synth.make_state_dict(1337).items() if k.startswith(('vfeat_extractor.', 'afeat_extractor.'))}
tr = AVCLIPTrainer(sd, dev)
M = n * 1569
qkv = (torch.randn(M, 2304, device=dev) * 0.5).bfloat16()
dO = (torch.randn(M, 768, device=dev) * 0.1).bfloat16()


def timeit(fn, iters=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for kind in ('time', 'space'):
    print(f'divided attention backward ({kind}), {n} segments: {timeit(lambda: tr._divided_bwd(qkv, dO, n, kind)):.2f} ms')
x = (torch.randn(M, 768, device=dev)).bfloat16()
dy = (torch.randn(M, 2304, device=dev) * 0.1).bfloat16()
print(f'linear backward 768 -> 2304 (qkv): {timeit(lambda: tr._lin_bwd("vfeat_extractor.blocks.0.attn.qkv", dy, x, M, tag="h")):.2f} ms')
dy2 = (torch.randn(M, 768, device=dev) * 0.1).bfloat16()
act = (torch.randn(M, 3072, device=dev)).bfloat16()
print(f'linear backward 3072 -> 768 (fc2): {timeit(lambda: tr._lin_bwd("vfeat_extractor.blocks.0.mlp.fc2", dy2, act, M, tag="act")):.2f} ms')
