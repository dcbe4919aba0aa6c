"""Host-side step right after the hot path for the inference configs: the offset class grid and the top-k read-out of the 21-way logits.

Mirrors (behaviour, not code) `make_class_grid` / `quantize_offset` (dataset/transforms.py:221-239) and the read-out of
`decode_single_video_prediction` (example.py:38-56).  Pure tensor arithmetic on a handful of numbers - it stays on the host; the logits come
from `Synchformer.forward` / `SynchformerEngine.forward` (HIP path).
"""
from typing import List, Optional, Tuple

import torch


def class_grid(leftmost: float, rightmost: float, n_classes: int, *, add_extreme_offset: bool = False, seg_size_vframes: Optional[int] = None,
               n_segments: Optional[int] = None, step_size_seg: Optional[float] = None, vfps: Optional[float] = None) -> torch.Tensor:
    """Offsets (seconds) the classes stand for: n_classes evenly spaced values on [leftmost, rightmost] in fp32 (configs/sync.yaml: 21 classes
    on [-2, 2] -> 0.2 s steps).  With `add_extreme_offset` one more class is appended: the length of the trimmed clip,
    (n_segments - (1 - step_size_seg) * (n_segments - 1)) * seg_size_vframes / vfps  (transforms.py:226-231)."""
    if n_classes < 3:
        raise ValueError(f'a grid of {n_classes} classes does not make sense (need >= 3)')
    # the reference builds the grid with numpy.linspace in float64 and casts to fp32: i * step + start with the end point set exactly
    step = (float(rightmost) - float(leftmost)) / (n_classes - 1)
    g64 = torch.arange(n_classes, dtype=torch.float64) * step + float(leftmost)
    g64[-1] = float(rightmost)
    grid = g64.float()
    if add_extreme_offset:
        if not all([seg_size_vframes, n_segments, step_size_seg, vfps]):
            raise ValueError('add_extreme_offset needs seg_size_vframes, n_segments, step_size_seg and vfps')
        trim_size_in_seg = n_segments - (1 - step_size_seg) * (n_segments - 1)
        grid = torch.cat([grid, torch.tensor([trim_size_in_seg * seg_size_vframes / vfps], dtype=torch.float32)])
    return grid


def quantize_offset(grid: torch.Tensor, off_sec: float) -> Tuple[float, int]:
    """Snap an offset in seconds onto the closest grid element -> (grid value, class index); the first closest element wins, like argmin."""
    idx = int((grid - off_sec).abs().argmin())
    return float(grid[idx]), idx


def topk_offsets(off_logits: torch.Tensor, grid: torch.Tensor, k: int = 5) -> List[List[Tuple[float, float, float, int]]]:
    """Per clip, the k most likely classes as (probability, logit, offset in seconds, class index), most likely first - the numbers
    `decode_single_video_prediction` prints (it handles one clip; this takes any batch)."""
    logits = off_logits.detach().float().cpu()
    if logits.dim() != 2 or logits.shape[-1] != grid.numel():
        raise ValueError(f'expected (B, {grid.numel()}) logits, got {tuple(logits.shape)}')
    probs = torch.softmax(logits, dim=-1)
    k = min(logits.shape[-1], k)
    top_logits, top_idx = torch.topk(logits, k)
    return [[(float(probs[b, i]), float(top_logits[b, j]), float(grid[i]), int(i)) for j, i in enumerate(top_idx[b])] for b in range(logits.shape[0])]


def offset_accuracy(targets: torch.Tensor, logits: torch.Tensor, topk=(1, 5)) -> dict:
    """The accuracy figures of the reference's `calc_cls_metrics` (scripts/train_utils.py:632-705) that define the benchmark's "Acc@1":
    `accuracy_k` = target among the k largest logits, and `accuracy_k_tol1` = target, target - 1 or target + 1 (clamped to the class range: an offset
    one 0.2 s grid step away counts, README.md:109-111) among the k largest.  (The mAP / ROC-AUC / d-prime part of that function is dataset
    statistics, not part of the path.)  targets (n,) int, logits (n, C) -> {'accuracy_1': .., 'accuracy_1_tol1': .., ...}."""
    logits = logits.detach().float().cpu()
    targets = targets.detach().long().cpu()
    if logits.dim() != 2 or targets.shape != logits.shape[:1]:
        raise ValueError(f'expected (n, C) logits and (n,) targets, got {tuple(logits.shape)} / {tuple(targets.shape)}')
    n, c = logits.shape
    ks = [min(k, c) for k in topk]
    preds = torch.topk(logits, k=max(ks), dim=1).indices                              # (n, kmax), best first (train_utils.py:665)
    t = targets.unsqueeze(-1).expand_as(preds)
    hit = preds == t
    hit_tol = hit | (preds == (t - 1).clamp(0, c - 1)) | (preds == (t + 1).clamp(0, c - 1))   # (train_utils.py:694-698)
    out = {}
    for k in ks:
        out[f'accuracy_{k}'] = float(hit[:, :k].any(dim=1).sum()) / max(n, 1)
        out[f'accuracy_{k}_tol1'] = float(hit_tol[:, :k].any(dim=1).sum()) / max(n, 1)
    return out
