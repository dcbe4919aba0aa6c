"""Synthetic checkpoint FILES in the formats the reference's constructors read, rebuilt bit-exactly from seeds wherever they
are needed: by tests/golden/make_golden.py (which loads them with the REAL reference classes and stores what those end up holding) and by the
tests that load the same files through synchformer_amd.checkpoint.  Data generators only - no reference code."""
from pathlib import Path

import torch

from synchformer_amd import synth

S1_SEED, HF_SEED = 2024, 2025


def tower_states(seed, gain=2.0):
    sd = synth.make_state_dict(seed, gain=gain)
    vsd = {k[len('vfeat_extractor.'):]: v for k, v in sd.items() if k.startswith('vfeat_extractor.')}
    asd = {k[len('afeat_extractor.'):]: v for k, v in sd.items() if k.startswith('afeat_extractor.')}
    return sd, vsd, asd


def write_stage1_ckpt(path: Path, seed: int = S1_SEED, gain: float = 2.0):
    """A Stage-1 AVCLIP `epoch_best.pt` as training/train.py saves it from the DDP wrapper: `state_dict` with `module.v_encoder.` / `module.a_encoder.`
    prefixes, plus keys the Stage-2 extractors do not have (the pooling time aggregator has no weights; a transformer one and the logit scale do)."""
    _, vsd, asd = tower_states(seed, gain)
    s1 = {'module.v_encoder.' + k: v for k, v in vsd.items()}
    s1.update({'module.a_encoder.' + k: v for k, v in asd.items()})
    s1['module.v_encoder.temp_attn_agg.cls_token'] = torch.full((1, 1, 768), 0.25)
    s1['module.a_encoder.temp_attn_agg.cls_token'] = torch.full((1, 1, 768), -0.25)
    s1['module.logit_scale'] = torch.tensor(0.05)
    torch.save({'state_dict': s1, 'epoch': 1}, path)
    return path


def hf_ast_state():
    """`ASTForAudioClassification.state_dict()` of MIT/ast-finetuned-audioset-10-10-0.4593 in shape: `audio_spectrogram_transformer.*` with the 1214-row
    position table the AudioSet model was trained with (12 x 101 patches + 2), and the 527-way classifier."""
    _, _, asd = tower_states(HF_SEED)
    hf = {'audio_spectrogram_transformer.' + k[len('ast.'):]: v for k, v in asd.items() if k.startswith('ast.')}
    g = torch.Generator().manual_seed(HF_SEED)
    hf['audio_spectrogram_transformer.embeddings.position_embeddings'] = torch.randn(1, 1214, 768, generator=g)
    hf['classifier.layernorm.weight'], hf['classifier.layernorm.bias'] = torch.ones(768), torch.zeros(768)
    hf['classifier.dense.weight'], hf['classifier.dense.bias'] = torch.zeros(527, 768), torch.zeros(527)
    return hf


def write_hf_ast_dir(path: Path):
    path.mkdir(parents=True, exist_ok=True)
    torch.save(hf_ast_state(), path / 'pytorch_model.bin')
    return path


def digest(sd):
    """name -> (sum, sum of |x|, first element) in float64: what the golden file stores per tensor."""
    import numpy as np
    names = sorted(sd)
    vals = np.array([[float(sd[k].double().sum()), float(sd[k].double().abs().sum()), float(sd[k].reshape(-1)[0])] for k in names], dtype=np.float64)
    return names, vals
