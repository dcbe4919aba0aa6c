"""BASELINE configs[4]'s Acc@1-parity number: the 32 structured 13-segment clips of tests/golden/syncability_logits_32.npz through the 2-way
synchronizability head (GlobalTransformerWithSyncabilityHead, sync_model.py:176-190) on the bf16 engine and on the MXFP8 towers, against the REAL reference's logits,
at the reference-like init and at a trained scale (run on the GPU box).   python tools/syncability_parity.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from synchformer_amd import synth
from synchformer_amd.engine import SynchformerEngine


def stats(got, ref):
    """d = l1 - l0 is what the 2-way decision reads.  `centred`: the decision boundary moved to the MEDIAN of the reference's d (a bias shift of the head; on these
    synthetic clips every d has one sign, so the plain argmax agrees trivially) - agreement then asks for the same side of the median; a flip counts as a tie when the
    reference's d is within 2 x max |dd| of the boundary."""
    d, r = (got[:, 1] - got[:, 0]).double(), (ref[:, 1] - ref[:, 0]).double()
    err = (d - r).abs()
    med = r.median()
    same = (d > med) == (r > med)
    ties = (~same) & ((r - med).abs() <= 2 * err.max())
    return dict(max_dd=float(err.max()), rms_dd=float(err.pow(2).mean().sqrt()), spread=float(r.max() - r.min()), std=float(r.std()),
                argmax=int((got.argmax(1) == ref.argmax(1)).sum()), centred=int(same.sum()), centred_flips_that_are_ties=int(ties.sum()),
                rank_corr=float(np.corrcoef(np.argsort(np.argsort(d.numpy())), np.argsort(np.argsort(r.numpy())))[0, 1]), max_logit_err=float((got - ref).abs().max()))


def main():
    g = np.load(Path(__file__).resolve().parent.parent / 'tests' / 'golden' / 'syncability_logits_32.npz')
    n, seed = int(g['n_clips']), int(g['seed'])
    dev = torch.device('cuda:0')
    for variant in ('gain1', 'trained'):
        if variant == 'gain1':
            sd = synth.make_state_dict(seed, n_pos=184, n_out=2, head='sync_head')
        else:
            sd = synth.make_state_dict(seed, gain=2.0, n_pos=184, n_out=2, head='sync_head')
            sd['transformer.sync_head.weight'] = sd['transformer.sync_head.weight'] * float(g['head_scale'])
        ref = torch.from_numpy(g['logits_' + variant])
        for fp8 in (False, True):
            eng = SynchformerEngine(sd, dev, fp8_towers=fp8)
            got = []
            for c0 in range(0, n, 16):
                u8, aud = synth.make_structured_clips(c0, min(16, n - c0), 13, seed)
                got.append(eng.forward(u8.to(dev), aud.to(dev)).cpu())
            print(variant, 'mxfp8' if fp8 else 'bf16 ', {k: (round(v, 5) if isinstance(v, float) else v) for k, v in stats(torch.cat(got), ref).items()}, flush=True)
            del eng
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
