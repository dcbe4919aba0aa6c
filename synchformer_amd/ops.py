"""Tensor-level wrappers over the C ABI (one Python function per `sf_*` entry point).

torch is plumbing here: device memory (`Tensor.data_ptr()`), the current HIP stream and dtype bookkeeping.
All compute happens in libsynchformer_hip.so; there is no eager / CPU fallback - a CPU tensor raises.
Also registered as `torch.ops.synchformer.*` custom ops (see `register_torch_ops`) so the kernels are visible
to the dispatcher, as the reference's callers would expect of a PyTorch-ROCm extension.
"""
import ctypes as C
import os
import threading
from pathlib import Path
from typing import Optional, Sequence

import torch

from . import _lib

SF_F32, SF_BF16, SF_F16, SF_U8 = 0, 1, 2, 3
EPI_NONE, EPI_GELU = 0, 1
_DT = {torch.float32: SF_F32, torch.bfloat16: SF_BF16, torch.float16: SF_F16, torch.uint8: SF_U8}

RowMap = Optional[Sequence[int]]   # (n12, n2, sA, s1, s2, off) or None


def rowmap(n12: int, n2: int, sA: int, s1: int, s2: int, off: int):
    return (int(n12), int(n2), int(sA), int(s1), int(s2), int(off))


def _map(m: RowMap):
    if m is None:
        return None
    assert len(m) == 6
    return (C.c_int64 * 6)(*m)


def _dev(t: torch.Tensor, name: str) -> int:
    if not t.is_cuda:
        raise RuntimeError(f'{name}: expected a HIP device tensor, got {t.device} (no CPU fallback exists)')
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ld(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, f'expected a row-major 2-D view, got {tuple(t.shape)} / {t.stride()}'
    return t.stride(0)


def gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *, M: Optional[int] = None,
         residual: Optional[torch.Tensor] = None, gelu: bool = False, c_map: RowMap = None, r_map: RowMap = None):
    """out[cmap(m)] = act(a[m] @ w.T + bias) (+ residual[rmap(m)]).  a (>=M, K) bf16, w (N, K) bf16."""
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    M = a.shape[0] if M is None else M
    if w.dim() == 3:                                        # k-tile-major weight (K / 64, N, 64), see ktile_major_weight()
        K, N = w.shape[0] * 64, w.shape[1]
        assert w.shape[2] == 64 and w.is_contiguous()
        w_ld = 64
    else:
        N, K = w.shape
        w_ld = _ld(w)
    assert a.shape[1] == K
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= N
    if residual is not None:
        assert residual.dtype == torch.float32
    rc = _lib.load().sf_gemm_bf16(
        _dev(a, 'a'), _ld(a), _dev(w, 'w'), w_ld, _dev(bias, 'bias') if bias is not None else None,
        _dev(out, 'out'), _DT[out.dtype], _ld(out), _map(c_map),
        _dev(residual, 'residual') if residual is not None else None, _ld(residual) if residual is not None else 0,
        _map(r_map), EPI_GELU if gelu else EPI_NONE, M, N, K, _stream())
    _lib.check(rc, 'sf_gemm_bf16')
    return out


def mx_scale_planes(rows: int, K: int, device) -> torch.Tensor:
    """Stage-major E8M0 scale planes for an MXFP8 operand of `rows` x K: uint8 (K / 128, rows padded to whole 256-row tiles, 4).  The padding is what
    lets sf_gemm_mxfp8 fetch a tile's scales of one stage as one contiguous KiB (one LDS-DMA piece)."""
    return torch.zeros(K // 128, ((rows + 255) // 256) * 256, 4, device=device, dtype=torch.uint8)


def quantize_mxfp8(x: torch.Tensor, q: torch.Tensor, scales: torch.Tensor, rows: Optional[int] = None):
    """bf16 x (rows, K) -> q uint8 (rows, K) OCP e4m3 bytes + scales uint8 (K / 128, >= rows, 4): E8M0, one per 32 consecutive k, stage-major
    (scales[k // 128, r, (k // 32) % 4])."""
    assert x.dtype == torch.bfloat16 and q.dtype == torch.uint8 and scales.dtype == torch.uint8
    rows = x.shape[0] if rows is None else rows
    K = x.shape[1]
    assert q.shape[1] == K and scales.dim() == 3 and scales.shape[0] == K // 128 and scales.shape[1] >= rows and scales.shape[2] == 4 and scales.is_contiguous()
    rc = _lib.load().sf_quantize_mxfp8(_dev(x, 'x'), _ld(x), _dev(q, 'q'), _ld(q), _dev(scales, 'scales'), scales.stride(0), rows, K, _stream())
    _lib.check(rc, 'sf_quantize_mxfp8')
    return q, scales


def layernorm_mxfp8(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, q: torch.Tensor, scales: torch.Tensor, eps: float, rows: Optional[int] = None):
    """(q, scales) = quantize_mxfp8(bf16(LayerNorm(x))) in one pass: x fp32 (rows, 768) -> q uint8 (rows, 768), scales uint8 (6, >= rows, 4)."""
    assert x.dtype == torch.float32 and x.shape[1] == 768 and q.dtype == torch.uint8 and q.shape[1] == 768
    rows = x.shape[0] if rows is None else rows
    assert scales.dtype == torch.uint8 and scales.dim() == 3 and scales.shape[0] == 6 and scales.shape[1] >= rows and scales.is_contiguous()
    rc = _lib.load().sf_layernorm768_mxfp8(_dev(x, 'x'), _ld(x), _dev(gamma, 'gamma'), _dev(beta, 'beta'), _dev(q, 'q'), _ld(q), _dev(scales, 'scales'),
                                           scales.stride(0), rows, float(eps), _stream())
    _lib.check(rc, 'sf_layernorm768_mxfp8')
    return q, scales


def gemm_mxfp8(a_q: torch.Tensor, a_s: torch.Tensor, w_q: torch.Tensor, w_s: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, *,
               M: Optional[int] = None, residual: Optional[torch.Tensor] = None, gelu: bool = False, out_scales: Optional[torch.Tensor] = None):
    """out[m] = act(dq(a)[m] @ dq(w).T + bias) (+ residual[m]) on MXFP8 operands (see quantize_mxfp8).  With a uint8 `out` and `out_scales`
    (N / 128, >= M, 4) the result itself leaves as MXFP8."""
    assert (out.dtype == torch.uint8) == (out_scales is not None)
    assert a_q.dtype == torch.uint8 and w_q.dtype == torch.uint8 and a_s.dtype == torch.uint8 and w_s.dtype == torch.uint8
    M = a_q.shape[0] if M is None else M
    N, K = w_q.shape
    assert a_q.shape[1] == K
    assert a_s.dim() == 3 and w_s.dim() == 3 and a_s.shape[0] == K // 128 and w_s.shape[0] == K // 128 and a_s.is_contiguous() and w_s.is_contiguous()
    rc = _lib.load().sf_gemm_mxfp8(_dev(a_q, 'a_q'), _ld(a_q), _dev(a_s, 'a_s'), a_s.stride(0), _dev(w_q, 'w_q'), _ld(w_q), _dev(w_s, 'w_s'), w_s.stride(0),
                                   _dev(bias, 'bias') if bias is not None else None, _dev(out, 'out'), _DT[out.dtype], _ld(out),
                                   _dev(out_scales, 'out_scales') if out_scales is not None else None, out_scales.stride(0) if out_scales is not None else 0,
                                   _dev(residual, 'residual') if residual is not None else None, _ld(residual) if residual is not None else 0,
                                   EPI_GELU if gelu else EPI_NONE, M, N, K, _stream())
    _lib.check(rc, 'sf_gemm_mxfp8')
    return out


def gemm_mx_res_ln(a_q: torch.Tensor, a_s: torch.Tensor, w_q: torch.Tensor, w_s: torch.Tensor, bias: Optional[torch.Tensor], x: torch.Tensor, gamma: torch.Tensor,
                   beta: torch.Tensor, y_q: torch.Tensor, y_s: torch.Tensor, eps: float, *, M: Optional[int] = None, residual: Optional[torch.Tensor] = None):
    """gemm_res_ln on MXFP8 operands with an MXFP8 output: x[m] = dq(a)[m] @ dq(w).T + bias + residual[m] (fp32, in place when residual is None or x),
    (y_q, y_s) = quantize_mxfp8(bf16(LayerNorm(x[m]) * gamma + beta)).  (y_q, y_s) may be (a_q, a_s) when K == 768."""
    assert a_q.dtype == torch.uint8 and w_q.dtype == torch.uint8 and a_s.dtype == torch.uint8 and w_s.dtype == torch.uint8 and y_q.dtype == torch.uint8 and y_s.dtype == torch.uint8
    assert x.dtype == torch.float32 and x.shape[1] == 768 and y_q.shape[1] == 768
    M = a_q.shape[0] if M is None else M
    K = a_q.shape[1]
    assert w_q.shape == (768, K)
    assert a_s.dim() == 3 and w_s.dim() == 3 and y_s.dim() == 3 and a_s.shape[0] == K // 128 and w_s.shape[0] == K // 128 and y_s.shape[0] == 6
    assert a_s.is_contiguous() and w_s.is_contiguous() and y_s.is_contiguous()
    r = x if residual is None else residual
    assert r.dtype == torch.float32
    rc = _lib.load().sf_gemm_mx_res_ln768(_dev(a_q, 'a_q'), _ld(a_q), _dev(a_s, 'a_s'), a_s.stride(0), _dev(w_q, 'w_q'), _ld(w_q), _dev(w_s, 'w_s'), w_s.stride(0),
                                          _dev(bias, 'bias') if bias is not None else None, _dev(r, 'residual'), _ld(r), _dev(x, 'x'), _ld(x),
                                          _dev(gamma, 'gamma'), _dev(beta, 'beta'), float(eps), _dev(y_q, 'y_q'), _ld(y_q), _dev(y_s, 'y_s'), y_s.stride(0),
                                          M, K, _stream())
    _lib.check(rc, 'sf_gemm_mx_res_ln768')
    return x, y_q, y_s


def gemm_res_ln(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                y: torch.Tensor, eps: float, *, M: Optional[int] = None, residual: Optional[torch.Tensor] = None):
    """x[m] = a[m] @ w.T + bias + residual[m] (fp32, in place when residual is None or x), y[m] = LayerNorm(x[m]) * gamma + beta (bf16).
    a (>= M, K) bf16, w (768, K) bf16; y may be the buffer `a` lives in (each 128-row tile reads its A rows before it writes them)."""
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dtype == torch.float32 and y.dtype == torch.bfloat16
    K = a.shape[1]
    if w.dim() == 3:                                        # k-step-major weight (K / 32, 768, 32), see kmajor_weight()
        assert w.shape == (K // 32, 768, 32) and w.is_contiguous()
        w_ptr, ldw = _dev(w, 'w'), 32
    else:
        assert w.shape == (768, K)
        w_ptr, ldw = _dev(w, 'w'), _ld(w)
    assert x.shape[1] == 768 and y.shape[1] == 768
    M = a.shape[0] if M is None else M
    r = x if residual is None else residual
    assert r.dtype == torch.float32
    rc = _lib.load().sf_gemm_res_ln768(_dev(a, 'a'), _ld(a), w_ptr, ldw, _dev(bias, 'bias') if bias is not None else None,
                                       _dev(r, 'residual'), _ld(r), _dev(x, 'x'), _ld(x), _dev(gamma, 'gamma'), _dev(beta, 'beta'), float(eps),
                                       _dev(y, 'y'), _ld(y), M, K, _stream())
    _lib.check(rc, 'sf_gemm_res_ln768')
    return x, y


def ktile_major_weight(w: torch.Tensor) -> torch.Tensor:
    """(N, K) bf16 Linear weight -> (K / 64, N, 64) for the persistent 256 x 256 x 64 kernel of sf_gemm_bf16 (contiguous 32 KiB W tile per k-tile)."""
    N, K = w.shape
    return w.view(N, K // 64, 64).permute(1, 0, 2).contiguous()


def kmajor_weight(w: torch.Tensor) -> torch.Tensor:
    """(768, K) bf16 Linear weight -> (K / 32, 768, 32): the layout sf_gemm_res_ln768 streams fastest (one contiguous 48 KiB slice per 32-deep k-step
    instead of 768 half cache lines 2 K bytes apart).  Pure data movement, done once at weight-preparation time."""
    N, K = w.shape
    return w.view(N, K // 32, 32).permute(1, 0, 2).contiguous()


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out: torch.Tensor, eps: float, *,
              rows: Optional[int] = None, in_map: RowMap = None, out_map: RowMap = None, accumulate: bool = False):
    assert x.dtype == torch.float32 and x.shape[1] == 768 and out.shape[1] == 768
    rows = x.shape[0] if rows is None else rows
    rc = _lib.load().sf_layernorm768(_dev(x, 'x'), _ld(x), _map(in_map), _dev(gamma, 'gamma'), _dev(beta, 'beta'),
                                     _dev(out, 'out'), _DT[out.dtype], _ld(out), _map(out_map), int(accumulate), rows,
                                     float(eps), _stream())
    _lib.check(rc, 'sf_layernorm768')
    return out


def broadcast_rows(dst: torch.Tensor, table: torch.Tensor, n_seq: int, dst_seq_rows: int):
    assert dst.dtype == torch.float32 and table.dtype == torch.float32 and table.shape[1] == 768 and table.is_contiguous()
    rc = _lib.load().sf_broadcast_rows768(_dev(dst, 'dst'), _ld(dst), dst_seq_rows, _dev(table, 'table'), table.shape[0],
                                          n_seq, _stream())
    _lib.check(rc, 'sf_broadcast_rows768')
    return dst


def gather_rows(x: torch.Tensor, out: torch.Tensor, rows: int, in_map: RowMap = None):
    assert x.dtype == torch.float32
    rc = _lib.load().sf_gather_rows768(_dev(x, 'x'), _ld(x), _map(in_map), _dev(out, 'out'), _DT[out.dtype], _ld(out), rows,
                                       _stream())
    _lib.check(rc, 'sf_gather_rows768')
    return out


def im2col_video(vid: torch.Tensor, out: torch.Tensor):
    """vid (n_seg, 16, 3, 224, 224) contiguous of u8/f16/bf16/f32 -> out bf16 (n_seg*1568, 1536)."""
    assert vid.is_contiguous() and tuple(vid.shape[1:]) == (16, 3, 224, 224), tuple(vid.shape)
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.shape[1] == 1536
    rc = _lib.load().sf_im2col_video(_dev(vid, 'vid'), _DT[vid.dtype], _dev(out, 'out'), vid.shape[0], _stream())
    _lib.check(rc, 'sf_im2col_video')
    return out


def im2col_video_clips(vid: torch.Tensor, out: torch.Tensor, frame0: int, seg_stride: int, n_seg: int):
    """vid (n_clips, T, 3, 224, 224) u8|f16|bf16|f32 -> out bf16 (n_clips*n_seg*1568, 1536): segment (clip, s) = frames
    [frame0 + s*seg_stride, +16) of the clip, read in place (GenerateMultipleSegments on the device, dataset/transforms.py:450-451)."""
    assert vid.is_contiguous() and vid.dim() == 5 and tuple(vid.shape[2:]) == (3, 224, 224)
    rc = _lib.load().sf_im2col_video_clips(_dev(vid, 'vid'), _DT[vid.dtype], vid.shape[0], vid.shape[1], frame0, seg_stride, n_seg,
                                           _dev(out, 'out'), _stream())
    _lib.check(rc, 'sf_im2col_video_clips')
    return out


def im2col_video_tokens(vid: torch.Tensor, out: torch.Tensor, frame0: int = 0, seg_stride: int = 0, n_seg: int = 1):
    """im2col_video / im2col_video_clips into the TOKEN layout: out bf16 (segments * 1569, 1536), every segment's row 0 zero, its patches at rows 1 .. 1568."""
    assert vid.is_contiguous() and vid.dim() == 5 and tuple(vid.shape[2:]) == (3, 224, 224)
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.shape[1] == 1536 and out.shape[0] >= vid.shape[0] * n_seg * 1569
    rc = _lib.load().sf_im2col_video_tokens(_dev(vid, 'vid'), _DT[vid.dtype], vid.shape[0], vid.shape[1], frame0, seg_stride, n_seg, _dev(out, 'out'), _stream())
    _lib.check(rc, 'sf_im2col_video_tokens')
    return out


def im2col_spec(spec: torch.Tensor, out: torch.Tensor):
    """spec fp32 (n_seg, F, Ta) contiguous -> out bf16 (n_seg*nf*nt, 256)."""
    assert spec.is_contiguous() and spec.dtype == torch.float32 and spec.dim() == 3
    assert out.dtype == torch.bfloat16 and out.is_contiguous() and out.shape[1] == 256
    rc = _lib.load().sf_im2col_spec(_dev(spec, 'spec'), _dev(out, 'out'), spec.shape[0], spec.shape[1], spec.shape[2],
                                    _stream())
    _lib.check(rc, 'sf_im2col_spec')
    return out


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *, n_seq: int, seq_rows: int,
              n_groups: int, row0: int, group_stride: int, tok_stride: int, n_tok: int, cls_row: int, heads: int,
              head_dim: int, scale: float, key_keep: Optional[torch.Tensor] = None):
    """q/k/v: column-slice views (rows, heads*head_dim) of a packed bf16 projection; see include/synchformer_hip.h.
    key_keep: optional uint8 (rows,) token mask, 0 = that K/V row is masked out for every query."""
    assert q.dtype == k.dtype == v.dtype == out.dtype == torch.bfloat16
    assert _ld(q) == _ld(k) == _ld(v)
    if key_keep is None:
        rc = _lib.load().sf_attention(_dev(q, 'q'), _dev(k, 'k'), _dev(v, 'v'), _ld(q), _dev(out, 'out'), _ld(out), n_seq,
                                      seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim,
                                      float(scale), _stream())
    else:
        assert key_keep.dtype == torch.uint8 and key_keep.is_contiguous()
        rc = _lib.load().sf_attention_masked(_dev(q, 'q'), _dev(k, 'k'), _dev(v, 'v'), _ld(q), _dev(out, 'out'), _ld(out), n_seq,
                                             seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim,
                                             float(scale), _dev(key_keep, 'key_keep'), _stream())
    _lib.check(rc, 'sf_attention')
    return out


def attention_cls_partial(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, partials: torch.Tensor, *, n_seq: int,
                          seq_rows: int, n_groups: int, row0: int, group_stride: int, tok_stride: int, n_tok: int, cls_row: int, heads: int,
                          head_dim: int, scale: float, key_keep: Optional[torch.Tensor] = None):
    """`attention` + per-group partials of the CLS query into `partials` (fp32, >= n_seq*heads*n_groups*66 elements).  key_keep: optional uint8 flags per
    K/V row (0 = masked key), as in `attention`."""
    assert q.dtype == k.dtype == v.dtype == out.dtype == torch.bfloat16 and partials.dtype == torch.float32
    assert _ld(q) == _ld(k) == _ld(v) and partials.numel() >= n_seq * heads * n_groups * 66
    if key_keep is not None:
        assert key_keep.dtype == torch.uint8 and key_keep.numel() >= n_seq * seq_rows
        rc = _lib.load().sf_attention_cls_partial_masked(_dev(q, 'q'), _dev(k, 'k'), _dev(v, 'v'), _ld(q), _dev(out, 'out'), _ld(out), n_seq, seq_rows,
                                                         n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim, float(scale),
                                                         _dev(partials, 'partials'), _dev(key_keep, 'key_keep'), _stream())
        _lib.check(rc, 'sf_attention_cls_partial_masked')
        return out
    rc = _lib.load().sf_attention_cls_partial(_dev(q, 'q'), _dev(k, 'k'), _dev(v, 'v'), _ld(q), _dev(out, 'out'), _ld(out), n_seq, seq_rows,
                                              n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim, float(scale),
                                              _dev(partials, 'partials'), _stream())
    _lib.check(rc, 'sf_attention_cls_partial')
    return out


def qkv_time_attention(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], qkv_cls: torch.Tensor, out: torch.Tensor, partials: torch.Tensor, *,
                       n_seq: int, n_groups: int, scale: float, key_keep: Optional[torch.Tensor] = None):
    """Temporal qkv projection + time attention of every patch token in one launch (sf_qkv_time_attention): x (n_seq * (1 + 8 n_groups), 768) bf16,
    w (2304, 768) bf16, qkv_cls (n_seq, 2304) bf16 = the projection of the CLS rows; out: patch rows of the attention output, partials: the CLS
    query's softmax partials, one per 4 patches, for attention_cls_combine(n_part=n_groups // 4)."""
    assert x.dtype == w.dtype == qkv_cls.dtype == out.dtype == torch.bfloat16 and partials.dtype == torch.float32
    assert x.shape[1] == 768 and tuple(w.shape) == (2304, 768) and qkv_cls.shape[0] >= n_seq and qkv_cls.shape[1] == 2304 and out.shape[1] == 768
    assert x.shape[0] >= n_seq * (1 + 8 * n_groups) and out.shape[0] >= n_seq * (1 + 8 * n_groups) and partials.numel() >= n_seq * 12 * (n_groups // 4) * 66 and n_groups % 4 == 0
    if key_keep is not None:                                   # token masks: one uint8 per row of x, 0 = masked key
        assert key_keep.dtype == torch.uint8 and key_keep.numel() >= n_seq * (1 + 8 * n_groups)
        rc = _lib.load().sf_qkv_time_attention_masked(_dev(x, 'x'), _ld(x), _dev(w, 'w'), _ld(w), _dev(bias, 'bias') if bias is not None else None,
                                                      _dev(qkv_cls, 'qkv_cls'), _ld(qkv_cls), _dev(out, 'out'), _ld(out), _dev(partials, 'partials'), n_seq,
                                                      n_groups, float(scale), _dev(key_keep, 'key_keep'), _stream())
        _lib.check(rc, 'sf_qkv_time_attention_masked')
        return out
    rc = _lib.load().sf_qkv_time_attention(_dev(x, 'x'), _ld(x), _dev(w, 'w'), _ld(w), _dev(bias, 'bias') if bias is not None else None,
                                           _dev(qkv_cls, 'qkv_cls'), _ld(qkv_cls), _dev(out, 'out'), _ld(out), _dev(partials, 'partials'), n_seq, n_groups,
                                           float(scale), _stream())
    _lib.check(rc, 'sf_qkv_time_attention')
    return out


def copy_rows_bf16(src: torch.Tensor, dst: torch.Tensor, rows: int, cols: int, src_map: RowMap = None, dst_map: RowMap = None):
    """dst[dst_map(r), :cols] = src[src_map(r), :cols] for r < rows (bf16, cols % 8 == 0; sf_copy_rows_bf16)."""
    assert src.dtype == dst.dtype == torch.bfloat16
    rc = _lib.load().sf_copy_rows_bf16(_dev(src, 'src'), _ld(src), _map(src_map), _dev(dst, 'dst'), _ld(dst), _map(dst_map), rows, cols, _stream())
    _lib.check(rc, 'sf_copy_rows_bf16')
    return dst


def space_side_rows(x: torch.Tensor, side_in: torch.Tensor, n_seq: int, seq_rows: int = 1569, n_tok: int = 196):
    """The rows qkv_space_attention / qkv_time_attention2 do NOT project themselves, gathered for one small GEMM: per sequence [the CLS row; for frame f its tokens 192 .. 195]
    -> side_in (n_seq * 33, 768) bf16 (row seq * 33, rows seq * 33 + 1 + 4 f + i).  One launch (sf_side_rows)."""
    assert x.dtype == side_in.dtype == torch.bfloat16 and x.shape[1] == side_in.shape[1] == 768 and seq_rows == 1 + 8 * n_tok
    assert x.shape[0] >= n_seq * seq_rows and side_in.shape[0] >= n_seq * 33
    rc = _lib.load().sf_side_rows(_dev(x, 'x'), _ld(x) * 2, _dev(side_in, 'side_in'), _ld(side_in) * 2, 1536, None, 0, None, 0, 0, n_seq, n_tok, _stream())
    _lib.check(rc, 'sf_side_rows')
    return side_in


def space_side_rows_mx(x_q: torch.Tensor, x_s: torch.Tensor, side_q: torch.Tensor, side_s: torch.Tensor, n_seq: int, n_tok: int = 196):
    """space_side_rows of an MXFP8 operand: the e4m3 rows x_q (rows, 768) uint8 AND their E8M0 scale dwords x_s (6, >= rows, 4) -> side_q (n_seq * 33, 768), side_s
    (6, >= n_seq * 33, 4), in the same launch."""
    assert x_q.dtype == x_s.dtype == side_q.dtype == side_s.dtype == torch.uint8 and x_q.shape[1] == side_q.shape[1] == 768
    assert x_s.dim() == 3 and side_s.dim() == 3 and x_s.shape[0] == side_s.shape[0] == 6 and x_s.is_contiguous() and side_s.is_contiguous()
    rows = n_seq * (1 + 8 * n_tok)
    assert x_q.shape[0] >= rows and x_s.shape[1] >= rows and side_q.shape[0] >= n_seq * 33 and side_s.shape[1] >= n_seq * 33
    rc = _lib.load().sf_side_rows(_dev(x_q, 'x_q'), _ld(x_q), _dev(side_q, 'side_q'), _ld(side_q), 768, _dev(x_s, 'x_s'), x_s.stride(0), _dev(side_s, 'side_s'),
                                  side_s.stride(0), 6, n_seq, n_tok, _stream())
    _lib.check(rc, 'sf_side_rows')
    return side_q, side_s


def qkv_space_attention(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], side: torch.Tensor, out: torch.Tensor, partials: torch.Tensor, *, n_seq: int,
                        scale: float, n_tok: int = 196, key_keep: Optional[torch.Tensor] = None):
    """Spatial qkv projection + space attention of every patch token in one launch (sf_qkv_space_attention): x (n_seq * 1569, 768) bf16, w (2304, 768) bf16,
    side (n_seq * 33, 2304) bf16 = the projection of the rows of space_side_rows(); out: patch rows of the attention output (a buffer of its own), partials: the CLS
    query's softmax partials, one per frame, for attention_cls_combine(n_part=8).  key_keep (uint8, one flag per row of x; 0 = masked key): the token-mask form
    (sf_qkv_space_attention_masked)."""
    assert x.dtype == w.dtype == side.dtype == out.dtype == torch.bfloat16 and partials.dtype == torch.float32
    rows = n_seq * (1 + 8 * n_tok)
    assert x.shape[1] == 768 and tuple(w.shape) == (2304, 768) and side.shape[0] >= n_seq * 33 and side.shape[1] == 2304 and out.shape[1] == 768
    assert x.shape[0] >= rows and out.shape[0] >= rows and partials.numel() >= n_seq * 12 * 8 * 66 and x.data_ptr() != out.data_ptr()
    if key_keep is not None:
        assert key_keep.dtype == torch.uint8 and key_keep.numel() >= rows and key_keep.is_contiguous()
        rc = _lib.load().sf_qkv_space_attention_masked(_dev(x, 'x'), _ld(x), _dev(w, 'w'), _ld(w), _dev(bias, 'bias') if bias is not None else None, _dev(side, 'side'),
                                                       _ld(side), _dev(out, 'out'), _ld(out), _dev(partials, 'partials'), n_seq, n_tok, float(scale),
                                                       _dev(key_keep, 'key_keep'), _stream())
        _lib.check(rc, 'sf_qkv_space_attention_masked')
        return out
    rc = _lib.load().sf_qkv_space_attention(_dev(x, 'x'), _ld(x), _dev(w, 'w'), _ld(w), _dev(bias, 'bias') if bias is not None else None, _dev(side, 'side'), _ld(side),
                                            _dev(out, 'out'), _ld(out), _dev(partials, 'partials'), n_seq, n_tok, float(scale), _stream())
    _lib.check(rc, 'sf_qkv_space_attention')
    return out


def qkv_time_attention2(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], side: torch.Tensor, out: torch.Tensor, partials: torch.Tensor, *, n_seq: int,
                        scale: float, n_tok: int = 196, key_keep: Optional[torch.Tensor] = None):
    """Temporal qkv projection + time attention of every patch token in one launch on the 192 x 384 main loop (sf_qkv_time_attention2): x (n_seq * 1569, 768) bf16,
    w (2304, 768) bf16, side (n_seq * 33, 2304) bf16 = the projection of the rows of space_side_rows(); out: patch rows of the attention output (a buffer of its own),
    partials: the CLS query's softmax partials, 33 per sequence and head, for attention_cls_combine(n_part=33)."""
    assert x.dtype == w.dtype == side.dtype == out.dtype == torch.bfloat16 and partials.dtype == torch.float32
    rows = n_seq * (1 + 8 * n_tok)
    assert x.shape[1] == 768 and tuple(w.shape) == (2304, 768) and side.shape[0] >= n_seq * 33 and side.shape[1] == 2304 and out.shape[1] == 768
    assert x.shape[0] >= rows and out.shape[0] >= rows and partials.numel() >= n_seq * 12 * 33 * 66 and x.data_ptr() != out.data_ptr()
    if key_keep is not None:                                # token masks (sf_qkv_time_attention2_masked): uint8, one flag per row of x, 0 = masked key
        assert key_keep.dtype == torch.uint8 and key_keep.numel() >= rows and key_keep.is_contiguous()
        rc = _lib.load().sf_qkv_time_attention2_masked(_dev(x, 'x'), _ld(x), _dev(w, 'w'), _ld(w), _dev(bias, 'bias') if bias is not None else None, _dev(side, 'side'),
                                                       _ld(side), _dev(out, 'out'), _ld(out), _dev(partials, 'partials'), n_seq, n_tok, float(scale),
                                                       _dev(key_keep, 'key_keep'), _stream())
        _lib.check(rc, 'sf_qkv_time_attention2_masked')
        return out
    rc = _lib.load().sf_qkv_time_attention2(_dev(x, 'x'), _ld(x), _dev(w, 'w'), _ld(w), _dev(bias, 'bias') if bias is not None else None, _dev(side, 'side'), _ld(side),
                                            _dev(out, 'out'), _ld(out), _dev(partials, 'partials'), n_seq, n_tok, float(scale), _stream())
    _lib.check(rc, 'sf_qkv_time_attention2')
    return out


def qkv_space_attention_mx(x_q: torch.Tensor, x_s: torch.Tensor, w_q: torch.Tensor, w_s: torch.Tensor, bias: Optional[torch.Tensor], side: torch.Tensor, out: torch.Tensor,
                           partials: torch.Tensor, *, n_seq: int, scale: float, out_scales: Optional[torch.Tensor] = None, n_tok: int = 196):
    """qkv_space_attention on MXFP8 operands: x_q (n_seq * 1569, 768) uint8 e4m3 + x_s (6, >= rows, 4) scale planes, w_q (2304, 768) + w_s (6, >= 2304, 4); side
    (n_seq * 33, 2304) bf16.  With a uint8 `out` and `out_scales` (6, >= rows, 4) the patch rows are written as MXFP8 (= quantize_mxfp8 of the bf16 output; buffers of
    their own, not x_q / x_s), else `out` is bf16."""
    assert x_q.dtype == w_q.dtype == x_s.dtype == w_s.dtype == torch.uint8 and side.dtype == torch.bfloat16 and partials.dtype == torch.float32
    assert (out.dtype == torch.uint8) == (out_scales is not None) and out.dtype in (torch.uint8, torch.bfloat16)
    rows = n_seq * (1 + 8 * n_tok)
    assert x_q.shape[1] == 768 and tuple(w_q.shape) == (2304, 768) and side.shape[0] >= n_seq * 33 and side.shape[1] == 2304 and out.shape[1] == 768
    assert x_q.shape[0] >= rows and out.shape[0] >= rows and x_s.dim() == 3 and w_s.dim() == 3 and x_s.shape[0] == 6 and w_s.shape[0] == 6
    assert x_s.shape[1] >= rows and w_s.shape[1] >= 2304 and x_s.is_contiguous() and w_s.is_contiguous() and partials.numel() >= n_seq * 12 * 8 * 66
    if out_scales is not None:
        assert out_scales.dtype == torch.uint8 and out_scales.dim() == 3 and out_scales.shape[0] == 6 and out_scales.shape[1] >= rows and out_scales.is_contiguous()
        assert out.data_ptr() != x_q.data_ptr() and out_scales.data_ptr() != x_s.data_ptr()
    rc = _lib.load().sf_qkv_space_attention_mx(_dev(x_q, 'x_q'), _ld(x_q), _dev(x_s, 'x_s'), x_s.stride(0), _dev(w_q, 'w_q'), _ld(w_q), _dev(w_s, 'w_s'), w_s.stride(0),
                                               _dev(bias, 'bias') if bias is not None else None, _dev(side, 'side'), _ld(side),
                                               _dev(out, 'out') if out_scales is None else None, _ld(out) if out_scales is None else 0,
                                               _dev(out, 'out') if out_scales is not None else None, _ld(out) if out_scales is not None else 0,
                                               _dev(out_scales, 'out_scales') if out_scales is not None else None, out_scales.stride(0) if out_scales is not None else 0,
                                               _dev(partials, 'partials'), n_seq, n_tok, float(scale), _stream())
    _lib.check(rc, 'sf_qkv_space_attention_mx')
    return out


def qkv_time_attention2_mx(x_q: torch.Tensor, x_s: torch.Tensor, w_q: torch.Tensor, w_s: torch.Tensor, bias: Optional[torch.Tensor], side: torch.Tensor, out: torch.Tensor,
                           partials: torch.Tensor, *, n_seq: int, scale: float, out_scales: Optional[torch.Tensor] = None, n_tok: int = 196):
    """qkv_time_attention2 on MXFP8 operands (arguments as qkv_space_attention_mx; partials: 33 records per sequence and head for attention_cls_combine(_mx)(n_part=33)).
    With a uint8 `out` and `out_scales` (6, >= rows, 4) the patch rows are written as MXFP8 (= quantize_mxfp8 of the bf16 output), else `out` is bf16."""
    assert x_q.dtype == w_q.dtype == x_s.dtype == w_s.dtype == torch.uint8 and side.dtype == torch.bfloat16 and partials.dtype == torch.float32
    assert (out.dtype == torch.uint8) == (out_scales is not None) and out.dtype in (torch.uint8, torch.bfloat16)
    rows = n_seq * (1 + 8 * n_tok)
    assert x_q.shape[1] == 768 and tuple(w_q.shape) == (2304, 768) and side.shape[0] >= n_seq * 33 and side.shape[1] == 2304 and out.shape[1] == 768
    assert x_q.shape[0] >= rows and out.shape[0] >= rows and x_s.dim() == 3 and w_s.dim() == 3 and x_s.shape[0] == 6 and w_s.shape[0] == 6
    assert x_s.shape[1] >= rows and w_s.shape[1] >= 2304 and x_s.is_contiguous() and w_s.is_contiguous() and partials.numel() >= n_seq * 12 * 33 * 66
    if out_scales is not None:
        assert out_scales.dtype == torch.uint8 and out_scales.dim() == 3 and out_scales.shape[0] == 6 and out_scales.shape[1] >= rows and out_scales.is_contiguous()
        assert out.data_ptr() != x_q.data_ptr() and out_scales.data_ptr() != x_s.data_ptr()
    rc = _lib.load().sf_qkv_time_attention2_mx(_dev(x_q, 'x_q'), _ld(x_q), _dev(x_s, 'x_s'), x_s.stride(0), _dev(w_q, 'w_q'), _ld(w_q), _dev(w_s, 'w_s'), w_s.stride(0),
                                               _dev(bias, 'bias') if bias is not None else None, _dev(side, 'side'), _ld(side),
                                               _dev(out, 'out') if out_scales is None else None, _ld(out) if out_scales is None else 0,
                                               _dev(out, 'out') if out_scales is not None else None, _ld(out) if out_scales is not None else 0,
                                               _dev(out_scales, 'out_scales') if out_scales is not None else None, out_scales.stride(0) if out_scales is not None else 0,
                                               _dev(partials, 'partials'), n_seq, n_tok, float(scale), _stream())
    _lib.check(rc, 'sf_qkv_time_attention2_mx')
    return out


def qkv_time_attention_mx(x_q: torch.Tensor, x_s: torch.Tensor, w_q: torch.Tensor, w_s: torch.Tensor, bias: Optional[torch.Tensor], qkv_cls: torch.Tensor,
                          out: torch.Tensor, partials: torch.Tensor, *, n_seq: int, n_groups: int, scale: float, out_scales: Optional[torch.Tensor] = None):
    """qkv_time_attention on MXFP8 operands: x_q (n_seq * (1 + 8 n_groups), 768) uint8 e4m3 + x_s (6, >= rows, 4) scale planes, w_q (2304, 768) + w_s (6, >= 2304, 4);
    qkv_cls (n_seq, 2304) bf16, out bf16, partials fp32 as in qkv_time_attention.  With a uint8 `out` and `out_scales` (6, >= rows, 4) the patch rows are written
    as MXFP8 (= quantize_mxfp8 of the bf16 output; buffers of their own, not x_q / x_s)."""
    assert x_q.dtype == w_q.dtype == x_s.dtype == w_s.dtype == torch.uint8 and qkv_cls.dtype == torch.bfloat16 and partials.dtype == torch.float32
    assert (out.dtype == torch.uint8) == (out_scales is not None) and out.dtype in (torch.uint8, torch.bfloat16)
    assert x_q.shape[1] == 768 and tuple(w_q.shape) == (2304, 768) and qkv_cls.shape[0] >= n_seq and qkv_cls.shape[1] == 2304 and out.shape[1] == 768
    rows = n_seq * (1 + 8 * n_groups)
    assert x_q.shape[0] >= rows and out.shape[0] >= rows and x_s.dim() == 3 and w_s.dim() == 3 and x_s.shape[0] == 6 and w_s.shape[0] == 6
    assert x_s.shape[1] >= rows and w_s.shape[1] >= 2304 and x_s.is_contiguous() and w_s.is_contiguous() and partials.numel() >= n_seq * 12 * (n_groups // 4) * 66
    if out_scales is not None:
        assert out_scales.dtype == torch.uint8 and out_scales.dim() == 3 and out_scales.shape[0] == 6 and out_scales.shape[1] >= rows and out_scales.is_contiguous()
        assert out.data_ptr() != x_q.data_ptr() and out_scales.data_ptr() != x_s.data_ptr()
        rc = _lib.load().sf_qkv_time_attention_mx_q(_dev(x_q, 'x_q'), _ld(x_q), _dev(x_s, 'x_s'), x_s.stride(0), _dev(w_q, 'w_q'), _ld(w_q), _dev(w_s, 'w_s'), w_s.stride(0),
                                                    _dev(bias, 'bias') if bias is not None else None, _dev(qkv_cls, 'qkv_cls'), _ld(qkv_cls), _dev(out, 'out'), _ld(out),
                                                    _dev(out_scales, 'out_scales'), out_scales.stride(0), _dev(partials, 'partials'), n_seq, n_groups, float(scale), _stream())
        _lib.check(rc, 'sf_qkv_time_attention_mx_q')
        return out
    rc = _lib.load().sf_qkv_time_attention_mx(_dev(x_q, 'x_q'), _ld(x_q), _dev(x_s, 'x_s'), x_s.stride(0), _dev(w_q, 'w_q'), _ld(w_q), _dev(w_s, 'w_s'), w_s.stride(0),
                                              _dev(bias, 'bias') if bias is not None else None, _dev(qkv_cls, 'qkv_cls'), _ld(qkv_cls), _dev(out, 'out'), _ld(out),
                                              _dev(partials, 'partials'), n_seq, n_groups, float(scale), _stream())
    _lib.check(rc, 'sf_qkv_time_attention_mx')
    return out


def attention_cls_partial_mx(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out_q: torch.Tensor, out_s: torch.Tensor, partials: torch.Tensor, *, n_seq: int,
                             seq_rows: int, n_groups: int, row0: int, group_stride: int, tok_stride: int, n_tok: int, cls_row: int, heads: int, scale: float):
    """`attention_cls_partial` (head_dim 64, 192 <= n_tok <= 207) writing MXFP8: out_q uint8 (rows, heads*64), out_s uint8 scale planes (heads*64/128, rows_padded, 4)
    as `mx_scale_planes` lays them out - byte for byte what `quantize_mxfp8` makes of the bf16 output."""
    assert q.dtype == k.dtype == v.dtype == torch.bfloat16 and out_q.dtype == out_s.dtype == torch.uint8 and partials.dtype == torch.float32
    assert _ld(q) == _ld(k) == _ld(v) and partials.numel() >= n_seq * heads * n_groups * 66 and out_s.dim() == 3 and out_s.shape[0] * 2 == heads
    rc = _lib.load().sf_attention_cls_partial_mx(_dev(q, 'q'), _dev(k, 'k'), _dev(v, 'v'), _ld(q), _dev(out_q, 'out_q'), _ld(out_q), _dev(out_s, 'out_s'), out_s.stride(0),
                                                 n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, float(scale),
                                                 _dev(partials, 'partials'), _stream())
    _lib.check(rc, 'sf_attention_cls_partial_mx')
    return out_q


def attention_cls_combine_mx(partials: torch.Tensor, out_q: torch.Tensor, out_s: torch.Tensor, *, n_part: int, n_seq: int, out_seq_rows: int, out_row: int, heads: int):
    """`attention_cls_combine` writing the CLS rows as MXFP8 into the buffers of `attention_cls_partial_mx`."""
    assert out_q.dtype == out_s.dtype == torch.uint8 and out_s.dim() == 3 and out_s.shape[0] * 2 == heads
    rc = _lib.load().sf_attention_cls_combine_mx(_dev(partials, 'partials'), n_part, _dev(out_q, 'out_q'), _ld(out_q), _dev(out_s, 'out_s'), out_s.stride(0), out_seq_rows,
                                                 out_row, n_seq, heads, _stream())
    _lib.check(rc, 'sf_attention_cls_combine_mx')
    return out_q


def attention_cls_combine(partials: torch.Tensor, out: torch.Tensor, *, n_part: int, n_seq: int, out_seq_rows: int, out_row: int, heads: int):
    rc = _lib.load().sf_attention_cls_combine(_dev(partials, 'partials'), n_part, _dev(out, 'out'), _ld(out), out_seq_rows, out_row, n_seq, heads,
                                              _stream())
    _lib.check(rc, 'sf_attention_cls_combine')
    return out


def attention_cls(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *, n_seq: int, q_seq_rows: int,
                  q_row: int, kv_seq_rows: int, kv_row0: int, n_keys: int, out_seq_rows: int, out_row: int, heads: int,
                  head_dim: int, scale: float, key_keep: Optional[torch.Tensor] = None):
    assert q.dtype == k.dtype == v.dtype == out.dtype == torch.bfloat16
    assert _ld(q) == _ld(k) == _ld(v)
    if key_keep is None:
        rc = _lib.load().sf_attention_cls(_dev(q, 'q'), q_seq_rows, q_row, _dev(k, 'k'), _dev(v, 'v'), _ld(q), kv_seq_rows,
                                          kv_row0, n_keys, _dev(out, 'out'), _ld(out), out_seq_rows, out_row, n_seq, heads,
                                          head_dim, float(scale), _stream())
    else:
        assert key_keep.dtype == torch.uint8 and key_keep.is_contiguous()
        rc = _lib.load().sf_attention_cls_masked(_dev(q, 'q'), q_seq_rows, q_row, _dev(k, 'k'), _dev(v, 'v'), _ld(q), kv_seq_rows,
                                                 kv_row0, n_keys, _dev(out, 'out'), _ld(out), out_seq_rows, out_row, n_seq, heads,
                                                 head_dim, float(scale), _dev(key_keep, 'key_keep'), _stream())
    _lib.check(rc, 'sf_attention_cls')
    return out


def token_mask_video(content_keep: torch.Tensor, w0_sign: torch.Tensor, out: torch.Tensor):
    """content_keep (n, 16, 3, 224, 224) bool|uint8 (True = kept) -> out uint8 (n*1569,) token keep flags (CLS kept)."""
    m = content_keep.contiguous().view(torch.uint8)
    rc = _lib.load().sf_token_mask_video(_dev(m, 'mask'), m.shape[0], _dev(w0_sign, 'w0_sign'), _dev(out, 'out'), _stream())
    _lib.check(rc, 'sf_token_mask_video')
    return out


def token_mask_spec(content_keep: torch.Tensor, w0_sign: torch.Tensor, out: torch.Tensor):
    """content_keep (n, F, Ta) bool|uint8 -> out uint8 (n*74,) token keep flags (CLS, DISTILL kept)."""
    m = content_keep.contiguous().view(torch.uint8)
    rc = _lib.load().sf_token_mask_spec(_dev(m, 'mask'), m.shape[0], m.shape[1], m.shape[2], _dev(w0_sign, 'w0_sign'), _dev(out, 'out'), _stream())
    _lib.check(rc, 'sf_token_mask_spec')
    return out


def meanpool_l2norm(x: torch.Tensor, out: torch.Tensor, t: int, normalize: bool):
    """x (n*t, 768) fp32 -> out (n, 768): mean over each run of t rows, then optional F.normalize (open_clip/model.py:530-531)."""
    assert x.dtype == torch.float32 and out.dtype == torch.float32 and x.shape[0] == out.shape[0] * t
    rc = _lib.load().sf_meanpool_l2norm768(_dev(x, 'x'), _ld(x), t, _dev(out, 'out'), _ld(out), int(bool(normalize)), out.shape[0], _stream())
    _lib.check(rc, 'sf_meanpool_l2norm768')
    return out


def similarity(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, scale: float):
    """out (n, m) = scale * a (n, d) @ b (m, d)^T in fp32 (open_clip/model.py:508-509)."""
    assert a.dtype == b.dtype == out.dtype == torch.float32 and a.shape[1] == b.shape[1]
    rc = _lib.load().sf_similarity_f32(_dev(a, 'a'), _ld(a), _dev(b, 'b'), _ld(b), _dev(out, 'out'), _ld(out), a.shape[0], b.shape[0],
                                       a.shape[1], float(scale), _stream())
    _lib.check(rc, 'sf_similarity_f32')
    return out


def cross_entropy(logits: torch.Tensor, targets: torch.Tensor, loss: torch.Tensor, dlogits: torch.Tensor = None):
    """loss[0] = F.cross_entropy(logits, targets) (mean) for fp32 (B, C) logits and int64 class targets; optionally
    dlogits = d loss / d logits in the same launch.  A target outside [0, C) gives a NaN loss (and a zero dlogits row)."""
    assert logits.dtype == torch.float32 and targets.dtype == torch.int64 and loss.dtype == torch.float32
    assert dlogits is None or (dlogits.dtype == torch.float32 and dlogits.shape == logits.shape)
    rc = _lib.load().sf_cross_entropy(_dev(logits, 'logits'), _ld(logits), _dev(targets, 'targets'), logits.shape[0], logits.shape[1],
                                      _dev(loss, 'loss'), _dev(dlogits, 'dlogits') if dlogits is not None else None,
                                      _ld(dlogits) if dlogits is not None else 0, 1.0, _stream())
    _lib.check(rc, 'sf_cross_entropy')
    return loss


# ----------------------------------------------------------------------------------------------------------------------
# PyTorch dispatcher registration (SURVEY §8b "custom-op contract"): the C-ABI launchers as `torch.ops.synchformer.*`
# out-variant custom ops (device_types = "cuda", i.e. HIP on ROCm).  They mutate their `out` argument and return nothing,
# which keeps them usable under torch.no_grad / autocast-free eager code and visible to the dispatcher; autograd for
# training goes through synchformer_amd.train.SyncTrainFunction, not through these leaf ops.
# ----------------------------------------------------------------------------------------------------------------------
_registered = False


# the operators libsynchformer_torch.so defines (csrc/sf_torch_library.cpp)
DISPATCHER_OPS = ('gemm_bf16', 'layernorm768', 'gemm_res_ln768', 'attention', 'attention_cls', 'attention_cls_partial', 'attention_cls_combine', 'im2col_video',
                  'qkv_time_attention', 'qkv_time_attention2', 'qkv_space_attention', 'qkv_time_attention2_masked', 'qkv_space_attention_masked', 'space_side_rows',
                  'space_side_rows_mx', 'quantize_mxfp8', 'layernorm768_mxfp8', 'gemm_mxfp8', 'gemm_mx_res_ln768', 'qkv_time_attention_mx', 'qkv_time_attention_mx_q',
                  'attention_cls_partial_mx', 'attention_cls_combine_mx', 'qkv_space_attention_mx', 'qkv_space_attention_mx_q', 'qkv_time_attention2_mx',
                  'qkv_time_attention2_mx_q')


def register_torch_ops():
    """Load the dispatcher library: `libsynchformer_torch.so` (csrc/sf_torch_library.cpp) DEFINES `torch.ops.synchformer.*` with TORCH_LIBRARY and implements
    them for the CUDA (= HIP on ROCm) dispatch key with TORCH_LIBRARY_IMPL, each operator one call into the C ABI on the current HIP stream.  Python adds what has no
    device code: the Meta / FakeTensor implementations (every op is an out-variant - it mutates its outputs and returns nothing -, so the abstract implementation only
    checks what the launcher would refuse) and the two functional ops with autograd (synchformer_amd/functional.py).  Raises if the library is not built - there is no
    Python-side fallback registration."""
    global _registered
    if _registered:
        return
    path = _lib.lib_path().parent / 'libsynchformer_torch.so'
    if not path.exists() and os.environ.get('SYNCHFORMER_HIP_LIB'):        # a measurement build given by path (tools/ab_*.sh): the dispatcher library of the package
        path = Path(__file__).resolve().parent / 'lib' / 'libsynchformer_torch.so'
    if not path.exists():
        raise RuntimeError(f'{path} not found: the dispatcher library is not built (python -c "import __graft_entry__ as g; g.build()")')
    _lib.load()                                                             # libsynchformer_hip.so first: the dispatcher library links against it
    torch.ops.load_library(str(path))

    def _fake(check=None):
        def impl(*args):
            if check is not None:
                check(*args)
            return None
        return impl

    def _chk_gemm(a, w, bias, out, residual, gelu):
        n = w.shape[1] if w.dim() == 3 else w.shape[0]
        torch._check(a.dim() == 2 and out.dim() == 2 and out.shape[0] == a.shape[0] and out.shape[1] == n, lambda: 'synchformer::gemm_bf16: out must be (M, N)')
        torch._check(a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16, lambda: 'synchformer::gemm_bf16: bf16 operands')
        torch._check(out.dtype in (torch.bfloat16, torch.float32), lambda: 'synchformer::gemm_bf16: out is bf16 or fp32')

    def _chk_ln(x, gamma, beta, out, eps):
        torch._check(x.shape[-1] == 768 and out.shape[-1] == 768 and x.dtype == torch.float32, lambda: 'synchformer::layernorm768: fp32 (rows, 768) in, 768 columns out')

    torch.library.register_fake('synchformer::gemm_bf16')(_fake(_chk_gemm))
    torch.library.register_fake('synchformer::layernorm768')(_fake(_chk_ln))
    for name in DISPATCHER_OPS:
        if name not in ('gemm_bf16', 'layernorm768'):
            torch.library.register_fake('synchformer::' + name)(_fake())

    from . import functional as _functional                                 # the functional ops with autograd (synchformer::linear, ::layer_norm768)
    _functional.register()
    _registered = True


class via_dispatcher:
    """Context manager: while active, the launches the engine's default schedule is made of go through the PyTorch dispatcher
    (`torch.ops.synchformer.*`, registered above) instead of straight into the C ABI - what a torch-side integration (profiler, dispatch modes,
    `torch.library` consumers) sees.  Calls with arguments the registered schemas do not carry (row maps, explicit M) fall through to the direct path.
        with ops.via_dispatcher(): logits = engine.forward(vis, aud)
    The results are the same launches on the same buffers (tests/test_e2e_gpu.py compares them bit for bit)."""
    NAMES = ('gemm', 'layernorm', 'gemm_res_ln', 'qkv_time_attention', 'attention_cls_partial', 'attention_cls_combine', 'quantize_mxfp8', 'layernorm_mxfp8',
             'gemm_mxfp8', 'gemm_mx_res_ln', 'qkv_time_attention_mx', 'attention_cls_partial_mx', 'attention_cls_combine_mx', 'qkv_time_attention2',
             'qkv_space_attention', 'qkv_space_attention_mx', 'space_side_rows', 'space_side_rows_mx', 'qkv_time_attention2_mx')

    _depth = 0                                                    # re-entrant: an inner `with` inside an active one changes nothing
    _lock = threading.RLock()                                     # the depth counter and the module-globals swap are one critical section
    calls_total = 0                                               # launches that went through torch.ops.synchformer.* since import (all instances)

    def __init__(self):
        self.calls = 0
        self._nested = False

    def __enter__(self):
        with via_dispatcher._lock:
            if via_dispatcher._depth > 0:
                via_dispatcher._depth += 1
                self._nested = True
                return self
            register_torch_ops()                                  # raises if libsynchformer_torch.so is missing: the route is then NOT marked active (every later
            self._swap_in()                                       # `with` raises again instead of silently taking the direct path)
            via_dispatcher._depth = 1
            return self

    def _swap_in(self):
        g = globals()
        o = self.orig = {n: g[n] for n in self.NAMES}
        t = torch.ops.synchformer

        def count(fn):
            def run(*a):
                self.calls += 1
                via_dispatcher.calls_total += 1
                return fn(*a)
            return run

        def gemm_(a, w, bias, out, *, M=None, residual=None, gelu=False, c_map=None, r_map=None):
            if M is not None or c_map is not None or r_map is not None:
                return o['gemm'](a, w, bias, out, M=M, residual=residual, gelu=gelu, c_map=c_map, r_map=r_map)
            count(t.gemm_bf16)(a, w, bias, out, residual, gelu)
            return out

        def layernorm_(x, gamma, beta, out, eps, **kw):
            if kw:
                return o['layernorm'](x, gamma, beta, out, eps, **kw)
            count(t.layernorm768)(x, gamma, beta, out, eps)
            return out

        def gemm_res_ln_(a, w, bias, x, gamma, beta, y, eps, *, M=None, residual=None):
            if M is not None or residual is not None:
                return o['gemm_res_ln'](a, w, bias, x, gamma, beta, y, eps, M=M, residual=residual)
            count(t.gemm_res_ln768)(a, w, bias, x, gamma, beta, y, eps)
            return x, y

        def qkv_time_(x, w, bias, qkv_cls, out, partials, *, n_seq, n_groups, scale, key_keep=None):
            count(t.qkv_time_attention)(x, w, bias, qkv_cls, out, partials, n_seq, n_groups, scale, key_keep)
            return out

        def qkv_time_mx_(x_q, x_s, w_q, w_s, bias, qkv_cls, out, partials, *, n_seq, n_groups, scale, out_scales=None):
            if out_scales is not None:
                count(t.qkv_time_attention_mx_q)(x_q, x_s, w_q, w_s, bias, qkv_cls, out, out_scales, partials, n_seq, n_groups, scale)
                return out
            count(t.qkv_time_attention_mx)(x_q, x_s, w_q, w_s, bias, qkv_cls, out, partials, n_seq, n_groups, scale)
            return out

        def attn_part_(q, k, v, out, partials, *, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim, scale, key_keep=None):
            count(t.attention_cls_partial)(q, k, v, out, partials, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, head_dim, scale,
                                           key_keep)
            return out

        def attn_comb_(partials, out, *, n_part, n_seq, out_seq_rows, out_row, heads):
            count(t.attention_cls_combine)(partials, out, n_part, n_seq, out_seq_rows, out_row, heads)
            return out

        def quant_(x, q, scales, rows=None):
            if rows is not None:
                return o['quantize_mxfp8'](x, q, scales, rows)
            count(t.quantize_mxfp8)(x, q, scales)
            return q, scales

        def ln_mx_(x, gamma, beta, q, scales, eps, rows=None):
            if rows is not None:
                return o['layernorm_mxfp8'](x, gamma, beta, q, scales, eps, rows)
            count(t.layernorm768_mxfp8)(x, gamma, beta, q, scales, eps)
            return q, scales

        def gemm_mx_(a_q, a_s, w_q, w_s, bias, out, *, M=None, residual=None, gelu=False, out_scales=None):
            if M is not None:
                return o['gemm_mxfp8'](a_q, a_s, w_q, w_s, bias, out, M=M, residual=residual, gelu=gelu, out_scales=out_scales)
            count(t.gemm_mxfp8)(a_q, a_s, w_q, w_s, bias, out, out_scales, residual, gelu)
            return out

        def gemm_mx_ln_(a_q, a_s, w_q, w_s, bias, x, gamma, beta, y_q, y_s, eps, *, M=None, residual=None):
            if M is not None or residual is not None:
                return o['gemm_mx_res_ln'](a_q, a_s, w_q, w_s, bias, x, gamma, beta, y_q, y_s, eps, M=M, residual=residual)
            count(t.gemm_mx_res_ln768)(a_q, a_s, w_q, w_s, bias, x, gamma, beta, y_q, y_s, eps)
            return x, y_q, y_s

        def attn_part_mx_(q, k, v, out_q, out_s, partials, *, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, scale):
            count(t.attention_cls_partial_mx)(q, k, v, out_q, out_s, partials, n_seq, seq_rows, n_groups, row0, group_stride, tok_stride, n_tok, cls_row, heads, scale)
            return out_q

        def attn_comb_mx_(partials, out_q, out_s, *, n_part, n_seq, out_seq_rows, out_row, heads):
            count(t.attention_cls_combine_mx)(partials, out_q, out_s, n_part, n_seq, out_seq_rows, out_row, heads)
            return out_q

        def qkv_time2_(x, w, bias, side, out, partials, *, n_seq, scale, n_tok=196, key_keep=None):
            if key_keep is not None:
                count(t.qkv_time_attention2_masked)(x, w, bias, side, out, partials, n_seq, scale, key_keep)
                return out
            count(t.qkv_time_attention2)(x, w, bias, side, out, partials, n_seq, scale)
            return out

        def qkv_space_(x, w, bias, side, out, partials, *, n_seq, scale, n_tok=196, key_keep=None):
            if key_keep is not None:
                count(t.qkv_space_attention_masked)(x, w, bias, side, out, partials, n_seq, scale, key_keep)
                return out
            count(t.qkv_space_attention)(x, w, bias, side, out, partials, n_seq, scale)
            return out

        def qkv_space_mx_(x_q, x_s, w_q, w_s, bias, side, out, partials, *, n_seq, scale, out_scales=None, n_tok=196):
            if out_scales is not None:
                count(t.qkv_space_attention_mx_q)(x_q, x_s, w_q, w_s, bias, side, out, out_scales, partials, n_seq, scale)
                return out
            count(t.qkv_space_attention_mx)(x_q, x_s, w_q, w_s, bias, side, out, partials, n_seq, scale)
            return out

        def side_rows_(x, out, n_seq):
            count(t.space_side_rows)(x, out, n_seq)
            return out

        def side_rows_mx_(x_q, x_s, side_q, side_s, n_seq):
            count(t.space_side_rows_mx)(x_q, x_s, side_q, side_s, n_seq)
            return side_q, side_s

        def qkv_time2_mx_(x_q, x_s, w_q, w_s, bias, side, out, partials, *, n_seq, scale, out_scales=None, n_tok=196):
            if out_scales is not None:
                count(t.qkv_time_attention2_mx_q)(x_q, x_s, w_q, w_s, bias, side, out, out_scales, partials, n_seq, scale)
                return out
            count(t.qkv_time_attention2_mx)(x_q, x_s, w_q, w_s, bias, side, out, partials, n_seq, scale)
            return out

        g.update(attention_cls_partial_mx=attn_part_mx_, attention_cls_combine_mx=attn_comb_mx_, qkv_time_attention2=qkv_time2_, qkv_space_attention=qkv_space_,
                 qkv_space_attention_mx=qkv_space_mx_, space_side_rows=side_rows_, space_side_rows_mx=side_rows_mx_, qkv_time_attention2_mx=qkv_time2_mx_)
        g.update(gemm=gemm_, layernorm=layernorm_, gemm_res_ln=gemm_res_ln_, qkv_time_attention=qkv_time_, attention_cls_partial=attn_part_,
                 attention_cls_combine=attn_comb_, quantize_mxfp8=quant_, layernorm_mxfp8=ln_mx_, gemm_mxfp8=gemm_mx_, gemm_mx_res_ln=gemm_mx_ln_,
                 qkv_time_attention_mx=qkv_time_mx_)

    def __exit__(self, *exc):
        with via_dispatcher._lock:
            via_dispatcher._depth -= 1
            if self._nested:
                self._nested = False
                return
            globals().update(self.orig)
