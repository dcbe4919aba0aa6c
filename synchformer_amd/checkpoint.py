"""Checkpoint / wire-format adapters: the four formats the reference's constructors read, turned into state
dicts for the drop-in classes.  Local files only - there is no network on the target boxes, and nothing is ever downloaded
(`check_if_file_exists_else_download`, utils/utils.py, is deliberately not reproduced): a missing file is a FileNotFoundError.

  * released Synchformer checkpoint        `ckpt['model']`, 513 keys                      (utils/logger.py:139-153, example.py:133-134)
  * Stage-1 AVCLIP checkpoint (`*.pt`)     `ckpt['state_dict']`, `[module.]v_encoder.` / `[module.]a_encoder.` prefixes
                                                                                          (motionformer.py:156-173, ast.py:113-131)
  * original Motionformer `*.pyth`         `ckpt['model_state']`                          (motionformer.py:52-58, 109-116)
  * HF `MIT/ast-finetuned-audioset-10-10-0.4593`  `audio_spectrogram_transformer.*` keys, 1214 position rows cut to the first
                                                  f*t + 2 = 74                            (ast.py:49-53, 240-245)
"""
import logging
from pathlib import Path
from typing import Dict, Mapping, Tuple

import torch

MFORMER_DIVIDED = 'ssv2_divided_224_16x4.pyth'
MFORMER_UNSUPPORTED = ('ssv2_motionformer_224_16x4.pyth', 'ssv2_joint_224_16x4.pyth')   # trajectory / joint attention: out of scope
HF_AST_NAME = 'MIT/ast-finetuned-audioset-10-10-0.4593'


def load_file(path) -> Mapping:
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f'{path}: checkpoint not found (nothing is downloaded - place the file there)')
    if path.suffix == '.safetensors':
        from safetensors.torch import load_file as st_load
        return st_load(str(path))
    return torch.load(str(path), map_location='cpu', weights_only=False)


def _strip_module(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}


def synchformer_state(ckpt: Mapping) -> Dict[str, torch.Tensor]:
    """`ckpt['model']` of a Stage-2 / released checkpoint (a bare state dict is accepted too)."""
    sd = ckpt['model'] if 'model' in ckpt and isinstance(ckpt['model'], Mapping) else ckpt
    return _strip_module(sd)


def stage1_tower_state(ckpt: Mapping, which: str) -> Dict[str, torch.Tensor]:
    """Keys of one tower out of a Stage-1 AVCLIP checkpoint: which = 'v_encoder' | 'a_encoder' (motionformer.py:156-163)."""
    sd = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    out = {}
    for k, v in sd.items():
        if k.startswith((f'module.{which}.', f'{which}.')):
            out[k.replace('module.', '').replace(f'{which}.', '', 1)] = v
    if not out:
        raise ValueError(f'no `{which}.` keys in the checkpoint')
    return out


def motionformer_pyth_state(ckpt: Mapping) -> Dict[str, torch.Tensor]:
    return dict(ckpt['model_state'])


def hf_ast_state(sd: Mapping[str, torch.Tensor], n_tokens: int = 74) -> Dict[str, torch.Tensor]:
    """HF ASTForAudioClassification weights -> the `ast.` sub-tree of the AST extractor; position rows [:f*t+2] (ast.py:240-245)."""
    out = {}
    for k, v in sd.items():
        if k.startswith('audio_spectrogram_transformer.'):
            k = 'ast.' + k[len('audio_spectrogram_transformer.'):]
            if k == 'ast.embeddings.position_embeddings':
                v = v[:, :n_tokens].clone()
            out[k] = v
    if not out:
        raise ValueError('no `audio_spectrogram_transformer.` keys: not an HF AST checkpoint')
    return out


def load_into(module: torch.nn.Module, sd: Mapping[str, torch.Tensor], what: str) -> Tuple[list, list]:
    """strict=False load with the reference's logging (motionformer.py:110-116): returns (missing, unexpected)."""
    own = module.state_dict()
    use = {k: v for k, v in sd.items() if k in own and tuple(own[k].shape) == tuple(v.shape)}
    shape_bad = [k for k, v in sd.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    if shape_bad:
        raise ValueError(f'{what}: shape mismatch for {shape_bad[:5]}')
    status = module.load_state_dict(use, strict=False)
    unexpected = [k for k in sd if k not in own]
    if status.missing_keys or unexpected:
        logging.warning(f'Loading exact {what} ckpt failed. Missing keys ({len(status.missing_keys)}): {status.missing_keys[:8]}..., '
                        f'Unexpected keys ({len(unexpected)}): {unexpected[:8]}...')
    else:
        logging.info(f'Loading {what} ckpt succeeded.')
    return list(status.missing_keys), unexpected


def init_motionformer(module: torch.nn.Module, ckpt_path: str):
    name = Path(ckpt_path).name
    if name in MFORMER_UNSUPPORTED:
        raise NotImplementedError(f'{name}: joint / trajectory attention checkpoints are outside the hot path (divided attention only)')
    ckpt = load_file(ckpt_path)
    if name == MFORMER_DIVIDED:
        return load_into(module, motionformer_pyth_state(ckpt), 'vfeat_extractor')
    if str(ckpt_path).endswith('.pt'):                                     # Stage-1 checkpoint (motionformer.py:63)
        return load_into(module, stage1_tower_state(ckpt, 'v_encoder'), 'vfeat_extractor')
    raise ValueError(f'ckpt_path {ckpt_path} is not supported.')          # motionformer.py:80


def init_ast(module: torch.nn.Module, ckpt_path: str):
    if str(ckpt_path).endswith('.pt'):                                     # Stage-1 checkpoint (ast.py:60, 113-131)
        return load_into(module, stage1_tower_state(load_file(ckpt_path), 'a_encoder'), 'afeat_extractor')
    p = Path(ckpt_path)
    if ckpt_path == HF_AST_NAME and not p.exists():
        raise FileNotFoundError(f'{HF_AST_NAME}: no network here - pass a local directory / file holding the HF weights '
                                '(pytorch_model.bin or model.safetensors) as ckpt_path instead')
    if p.is_dir():
        cands = [p / 'model.safetensors', p / 'pytorch_model.bin']
        found = [c for c in cands if c.exists()]
        if not found:
            raise FileNotFoundError(f'{p}: neither model.safetensors nor pytorch_model.bin inside')
        p = found[0]
    n_tok = module.state_dict()['ast.embeddings.position_embeddings'].shape[1]
    return load_into(module, hf_ast_state(load_file(p), n_tok), 'afeat_extractor')
