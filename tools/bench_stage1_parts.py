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

# ---- anatomy of the space-attention backward (generic gathered path) -------------------------------------------------
from synchformer_amd.stage1 import copy_rows, D, VIS_L, VIS_P, H, HD  # noqa: E402
G, T, Lg, tok, grp, cls_tok, cls_grp = tr._group_maps('space')
nseq, rows_g = n * G, n * G * Lg
Gq = tr._buf('g_qkv', (rows_g, 3 * D), torch.bfloat16)
GdO = tr._buf('g_dO', (rows_g, D), torch.bfloat16, zero=True)
Gd = tr._buf('g_dqkv', (rows_g, 3 * D), torch.bfloat16)
dqkv = tr._buf('dqkv', (M, 3 * D), torch.bfloat16)


def gather():
    copy_rows(qkv, Gq, n * VIS_P, 3 * D, tok, grp)
    copy_rows(qkv, Gq, nseq, 3 * D, cls_tok, cls_grp)
    GdO.zero_()
    copy_rows(dO, GdO, n * VIS_P, D, tok, grp)


print(f'  gather qkv/dO into group sequences : {timeit(gather):.3f} ms')
print(f'  attn_bwd_seq (5 batched GEMMs + softmax + 5 transposes): {timeit(lambda: tr._attn_bwd_chunked(Gq, GdO, Gd, nseq, Lg)):.3f} ms')
print(f'  scatter back                        : {timeit(lambda: copy_rows(Gd, dqkv, n * VIS_P, 3 * D, grp, tok)):.3f} ms')
print(f'  CLS-query backward                  : {timeit(lambda: tr._cls_bwd(qkv, dO, dqkv, n, VIS_L, VIS_L, do_seq_rows=VIS_L, accumulate=True)):.3f} ms')
