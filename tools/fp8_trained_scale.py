"""MXFP8 towers at TRAINED logit scale: both variants of tests/golden/logits_only_32.npz through fp8_towers=True, against the real reference's
fp32 logits and against the bf16 engine.    python tools/fp8_trained_scale.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from synchformer_amd import synth
from synchformer_amd.engine import SynchformerEngine
from synchformer_amd.postprocess import offset_accuracy

gpu = torch.device('cuda:0')
g = np.load(Path(__file__).resolve().parent.parent / 'tests/golden/logits_only_32.npz')
n = int(g['n_clips'])
for variant in ('gain1', 'trained'):
    sd = synth.make_state_dict(int(g['seed'])) if variant == 'gain1' else synth.make_state_dict(int(g['seed']), gain=2.0)
    if variant == 'trained':
        sd['transformer.off_head.weight'] = sd['transformer.off_head.weight'] * float(g['head_scale'])
    out = {}
    for fp8 in (False, True):
        eng = SynchformerEngine(sd, gpu, seg_chunk=224, fp8_towers=fp8)
        got = []
        for c0 in range(0, n, 16):
            u8, aud = synth.make_structured_clips(c0, min(16, n - c0), 14, int(g['seed']))
            got.append(eng.forward(u8.to(gpu), aud.to(gpu)).cpu())
        out[fp8] = torch.cat(got)
        del eng
        torch.cuda.empty_cache()
    ref = torch.from_numpy(g['logits_' + variant])
    rng = float(ref.max() - ref.min())
    for name, got in (('bf16', out[False]), ('mxfp8', out[True])):
        err = float((got - ref).abs().max())
        rms = float((got - ref).pow(2).mean().sqrt())
        picked = ref.gather(1, got.argmax(1, keepdim=True)).squeeze(1)
        gap = ref.max(1).values - picked
        acc = offset_accuracy(ref.argmax(1), got, topk=(1, 5))
        top2 = ref.topk(2, dim=1).values
        print(f'{variant:8s} {name:6s}: max|d| {err:.4f} = {100 * err / rng:.2f} % of range {rng:.2f}, rms {rms:.4f} | argmax agree {int((got.argmax(1) == ref.argmax(1)).sum())}/{n} '
              f'worst gap of a flip {float(gap.max()):.4f} (median top1-top2 margin of the reference {float((top2[:, 0] - top2[:, 1]).median()):.4f}) | {acc}')
    d = float((out[True] - out[False]).abs().max())
    print(f'{variant:8s} mxfp8 vs bf16 engine: max|d| {d:.4f} = {100 * d / rng:.2f} % of range')
