"""Stage-1 train step on the HIP path: segment-level audio-visual contrastive pre-training of BOTH feature extractors
(train_clip_src/open_clip/model.py:449-585, train_clip_src/training/train.py:72-160,
train_clip.py:268-278, configs/segment_avclip.yaml).

`AVCLIPTrainer` keeps every trainable tensor of the two towers (+ logit_scale) in ONE flat fp32 master buffer with flat grad /
Adam m, v / bf16 operand copies (FlatTrainer), runs the towers with saved activations (~26 MB per visual segment per block,
sized for 288 GB of HBM: the reference's batch of 2 x 14 segments needs ~20 GB), the hand-scheduled backward, the flat-bucket
gradient all-reduce and the fused clip + Adam step (AdamW with the config's weight_decay 0.0).

Backward of divided space-time attention (vit_helper.py:100-158): each patch attends [CLS; its time or space group], the CLS
query attends everything.  The group part is run as ordinary self-attention over GATHERED sequences [CLS; group] (bf16 row
gather, sf_copy_rows_bf16) through the five strided-batched GEMMs of FlatTrainer.attn_bwd_seq; the zero dO row in the CLS slot
makes the CLS "query" of a group inert.  dq|dk|dv rows are scattered back, the CLS key/value gradient is summed over groups
(sf_reduce_groups_bf16) and the CLS-query part is added by sf_attention_cls_bwd.

All compute is in libsynchformer_hip; torch provides memory, the stream, torch.distributed, and two host-side scalars
(the logit_scale clamp and its gradient).
"""
import math
import os
from typing import Dict, Optional

import torch

from . import _lib, ops, synth
from .engine import AGG_A, AGG_V, AUD_L, AUD_P, D, EPS_AST, EPS_VIS, FF, VIS_L, VIS_P
from .train import FlatTrainer, _chk, _st, cast_bf16, colsum, ln_bwd

V, A = 'vfeat_extractor', 'afeat_extractor'
H, HD = 12, 64


def _m(t):
    import ctypes as C
    return (C.c_int64 * 6)(*t) if t is not None else None


def copy_rows(src, dst, rows, cols, src_map=None, dst_map=None):
    _chk(_lib.load().sf_copy_rows_bf16(src.data_ptr(), src.stride(0), _m(src_map), dst.data_ptr(), dst.stride(0), _m(dst_map), rows, cols, _st()),
         'sf_copy_rows_bf16')


def normalise_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """AVCLIP checkpoints name the towers v_encoder / a_encoder (open_clip/model.py:460-461); Synchformer names them
    vfeat_extractor / afeat_extractor (sync_model.py:27-28).  Internally the Synchformer names are used."""
    out = {}
    for k, v in sd.items():
        if k.startswith('v_encoder.'):
            k = V + k[len('v_encoder'):]
        elif k.startswith('a_encoder.'):
            k = A + k[len('a_encoder'):]
        out[k] = v
    return out


def cosine_lr(step: int, base_lr: float, warmup: int, total_steps: int) -> float:
    """train_clip_src/training/scheduler.py:9-10, 43-53 (configs/segment_avclip.yaml: lr_scheduler cosine, warmup 1000)."""
    if step < warmup:
        return base_lr * (step + 1) / warmup
    return 0.5 * (1 + math.cos(math.pi * (step - warmup) / (total_steps - warmup))) * base_lr


class AVCLIPTrainer(FlatTrainer):
    def __init__(self, state_dict: Dict[str, torch.Tensor], device='cuda:0', lr: float = 1e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_clip_norm: float = 1.0, clamp_scale=(0.001, 0.5), init_scale: float = 0.07, gather_for_loss: bool = False,
                 drop_path_rate: float = 0.2, seed: int = 1337):
        sd = dict(normalise_keys(state_dict))
        if 'logit_scale' not in sd:
            sd['logit_scale'] = torch.tensor(init_scale)
        keys = [k for k in sd if k.startswith((V + '.', A + '.')) and not k.startswith(V + '.patch_embed.')] + ['logit_scale']
        self._init_flat(sd, keys, device, lr, betas, eps, max_clip_norm)
        self.clamp_scale, self.gather_for_loss = clamp_scale, gather_for_loss
        # stochastic depth of the visual tower: DROP_PATH 0.2 (divided_224_16x4.yaml:59), block i drops its space-attention and MLP branches per
        # segment with probability linspace(0, rate, depth)[i] (video_model_builder.py:86-87, vit_helper.py:356,372,375); 0 = evaluation mode
        self.drop_path_rate, self.seed, self.fwd_count = float(drop_path_rate or 0.0), int(seed), 0
        self.time_comm, self._comm_ev = False, None                            # bench.py: HIP events around the bucket waits
        self.fused_attn_bwd = True          # False: the gathered batched-GEMM attention backward (kept as a cross-check)
        # the CLS query's backward of the space attention inside sf_attention_group_bwd (as one more query row per group, on the forward's softmax statistics)
        # instead of sf_attention_cls_bwd's read-modify-write pass over dk | dv; SF_CLS_IN_GROUP=0 keeps the separate pass
        self.cls_in_group = os.environ.get('SF_CLS_IN_GROUP', '1') != '0'
        # the LayerNorm backward that closes a residual branch also writes the next branch's dY operand and output-bias gradient (sf_layernorm768_bwd_branch)
        self.fuse_branch = os.environ.get('SF_FUSE_BRANCH', '1') != '0'
        self._pre_dy = {}
        self.two_streams = os.environ.get('SF_STAGE1_TWO_STREAMS', '1') != '0'   # audio tower next to the visual one (forward_backward)
        self._side = None
        self._ls_host = self._ls_ev = None  # pinned host mirror of logit_scale + the event behind its copy (forward_backward / _head)
        self._ls_pending = False
        self._dp_ones = None                # the all-ones vector sf_dropout turns into DropPath scales (_dp_scales)
        self._pre_ln = set()                # workspace buffers whose LayerNorm output _add_branch has already produced (consumed by _ln_into)
        self.n_vblocks = len([k for k in keys if k.startswith(V + '.blocks.') and k.endswith('.norm1.weight')])
        self.n_alayers = len([k for k in keys if k.endswith('.layernorm_before.weight')])

    # ---- small launch helpers ---------------------------------------------------------------------------------------
    def _ln(self, name):
        return self.p[name + '.weight'], self.p[name + '.bias']

    def _lnws(self, rows):
        return self._buf('ln_ws', (3 * 768 * ((rows + 3) // 4),), torch.float32)

    def _ln_bwd(self, x, name, dy, dx, rows, eps, next_branch=None, **kw):
        """next_branch = (dp, seq_rows, bias_key) of the residual branch whose backward runs next on this dx: its dY operand and output-bias gradient are written by
        this launch (sf_layernorm768_bwd_branch) and _branch_dy finds them done."""
        if next_branch is not None and self.fuse_branch and kw.get('acc_dx') and not (set(kw) - {'acc_dx'}):
            dp, seq_rows, bias_key = next_branch
            dy_b = self._buf('dy_b', (rows, D), torch.bfloat16)
            _chk(_lib.load().sf_layernorm768_bwd_branch(x.data_ptr(), x.stride(0), self.p[name + '.weight'].data_ptr(), dy.data_ptr(),
                                                        1 if dy.dtype == torch.bfloat16 else 0, dy.stride(0), dx.data_ptr(), dx.stride(0), 1,
                                                        self.g[name + '.weight'].data_ptr(), self.g[name + '.bias'].data_ptr(), 0, dy_b.data_ptr(), dy_b.stride(0),
                                                        dp.data_ptr() if dp is not None else None, max(1, seq_rows), self.g[bias_key].data_ptr(),
                                                        self._lnws(rows).data_ptr(), rows, float(eps), _st()), 'sf_layernorm768_bwd_branch')
            self._pre_dy[self._ws_prefix] = (bias_key, rows)
            return
        ln_bwd(x, self.p[name + '.weight'], dy, dx, self.g[name + '.weight'], self.g[name + '.bias'], self._lnws(rows), rows, eps, **kw)

    def _gelu_fwd(self, pre, act):
        _chk(_lib.load().sf_gelu_fwd(pre.data_ptr(), act.data_ptr(), pre.numel(), _st()), 'sf_gelu_fwd')

    def _gelu_bwd(self, pre, dact, dpre):
        if dact.dtype == torch.bfloat16:
            _chk(_lib.load().sf_gelu_bwd_bf16(pre.data_ptr(), dact.data_ptr(), dpre.data_ptr(), pre.numel(), _st()), 'sf_gelu_bwd_bf16')
        else:
            _chk(_lib.load().sf_gelu_bwd(pre.data_ptr(), dact.data_ptr(), dpre.data_ptr(), pre.numel(), _st()), 'sf_gelu_bwd')

    def _seqsum(self, x, n_seq, L, out):
        _chk(_lib.load().sf_seqsum(x.data_ptr(), x.stride(0), n_seq, L, D, out.data_ptr(), 0, _st()), 'sf_seqsum')

    # ---- stochastic depth ---------------------------------------------------------------------------------------------------
    def _dp_scales(self, block: int, site: int, n: int):
        """Per-segment branch scales (0 or 1 / keep) of DropPath site `site` (0 = space attention, 1 = MLP) of visual block `block` for the current
        forward pass, or None when that site is inactive.  sf_dropout over a vector of ones: counter-based, so the backward needs no saved mask."""
        p_ = self.drop_path_rate * block / max(1, self.n_vblocks - 1)
        if p_ <= 0.0:
            return None
        h = (self.seed * 0x9E3779B1 + self.fwd_count * 0x85EBCA6B + (2 * block + site) * 0xC2B2AE35 + 0x27D4EB2F) & 0xFFFFFFFF
        h ^= h >> 15
        w = ((n + 3) // 4) * 4
        if self._dp_ones is None or self._dp_ones.shape[1] != w:              # a constant: written once, not once per site and step (22 fill launches)
            self._dp_ones = torch.ones(1, w, device=self.dev, dtype=torch.float32)
        ones = self._dp_ones
        sc = self._buf(f'dp_scale_{block}_{site}', (1, ((n + 3) // 4) * 4), torch.float32)
        from .train import dropout
        dropout(ones, sc, 1, ones.shape[1], p_, (h * 0x2C1B3C6D) & 0xFFFFFFFF)
        return sc

    def _scale_seq(self, x, scales, seq_rows, rows, out, residual=None):
        """out = (residual +) scales[row // seq_rows] * x  (fp32, 768 columns)."""
        _chk(_lib.load().sf_scale_seq_add(x.data_ptr(), x.stride(0), scales.data_ptr(), seq_rows, residual.data_ptr() if residual is not None else None,
                                          residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), rows, D, _st()),
             'sf_scale_seq_add')
        return out

    def _branch_dy(self, dx, dp, seq_rows, rows, bias_key):
        """Head of a residual branch's backward: dY = bf16(dp[segment] * dx) for the branch's output Linear, and that Linear's bias gradient (fp32 column
        sums of the scaled gradient) into g[bias_key], in one pass over dx (sf_branch_grad)."""
        dy_b = self._buf('dy_b', (rows, D), torch.bfloat16)
        if self._pre_dy.pop(self._ws_prefix, None) == (bias_key, rows):        # written by the LayerNorm backward that last updated dx
            return dy_b
        ws = self._buf('colsum_ws', (D * ((rows + 63) // 64),), torch.float32)
        _chk(_lib.load().sf_branch_grad(dx.data_ptr(), dx.stride(0), dp.data_ptr() if dp is not None else None, max(1, seq_rows), dy_b.data_ptr(), dy_b.stride(0),
                                        rows, D, self.g[bias_key].data_ptr(), 0, ws.data_ptr(), _st()), 'sf_branch_grad')
        return dy_b

    def _add_branch(self, br, dp, seq_rows, rows, x_res, out, next_ln=None):
        """out = x_res + dp[segment] * br; with next_ln = (norm name, buffer name, eps) the LayerNorm that reads `out` next is computed in the same pass
        (sf_add_scale_ln768) into that buffer, and the call site of that norm finds it done (_ln_into)."""
        if next_ln is None:
            return self._scale_seq(br, dp, seq_rows, rows, out, residual=x_res)
        name, bufname, eps = next_ln
        y = self._buf(bufname, (rows, D), torch.bfloat16)
        g, b = self._ln(name)
        _chk(_lib.load().sf_add_scale_ln768(br.data_ptr(), br.stride(0), dp.data_ptr(), seq_rows, x_res.data_ptr(), x_res.stride(0), out.data_ptr(), out.stride(0),
                                            g.data_ptr(), b.data_ptr(), y.data_ptr(), y.stride(0), rows, eps, _st()), 'sf_add_scale_ln768')
        self._pre_ln.add(self._ws_prefix + bufname)
        return out

    def _clear_pre_ln(self):
        """Forget the pre-computed LayerNorms of THIS tower's workspaces (the two towers run on two streams with their own `_ws_prefix`): after an
        interrupted forward a stale key would make a later _ln_into skip its LayerNorm and read the previous step's buffer."""
        pre = self._ws_prefix
        self._pre_ln = {k for k in self._pre_ln if (not k.startswith(pre) if pre else k.startswith('a:'))}

    def _ln_into(self, x, name, bufname, rows, eps):
        """LayerNorm `name` of x into the workspace buffer `bufname` (bf16) - unless _add_branch already produced it together with x."""
        y = self._buf(bufname, (rows, D), torch.bfloat16)
        key = self._ws_prefix + bufname
        if key in self._pre_ln:
            self._pre_ln.discard(key)
        else:
            ops.layernorm(x, *self._ln(name), y, eps)
        return y

    def _mlp_fwd(self, s, x_in, h_name, fc1, fc2, rows, eps_name, eps, tag, dp=None, seq_rows=0, next_ln=None):
        """h = LN(x_in); pre = fc1(h); act = gelu(pre); returns x_in + fc2(act) (x_in + dp[segment] * fc2(act) under stochastic depth).
        Saves h, pre, act.  next_ln: the norm that reads the result next (see _add_branch)."""
        s['h2'] = self._ln_into(x_in, eps_name, f'{tag}_h2', rows, eps)
        s['pre'] = self._buf(f'{tag}_pre', (rows, FF), torch.bfloat16)
        s['act'] = self._buf(f'{tag}_act', (rows, FF), torch.bfloat16)
        w1, b1 = self._wb(fc1)
        # fc1 + GELU with both the pre-activation and the activation kept, in one launch where the quadrant-phased kernel applies (the big visual MLPs)
        rc = _lib.SF_NOT_APPLICABLE if rows < 8192 else _lib.load().sf_gemm_bf16_gelu_dual(s['h2'].data_ptr(), s['h2'].stride(0), w1.data_ptr(), w1.stride(0), b1.data_ptr(),
                                                                       s['pre'].data_ptr(), s['act'].data_ptr(), FF, rows, FF, D, _st())
        if rc == _lib.SF_NOT_APPLICABLE:                       # shape outside config 11's range: nothing was launched; any other non-zero code is an error
            ops.gemm(s['h2'], w1, b1, s['pre'])
            self._gelu_fwd(s['pre'], s['act'])
        else:
            _chk(rc, 'sf_gemm_bf16_gelu_dual')
        out = self._buf(f'{tag}_xo', (rows, D), torch.float32)
        if dp is None:
            ops.gemm(s['act'], *self._wb(fc2), out, residual=x_in)
        else:
            br = self._buf('dp_branch', (rows, D), torch.float32)
            ops.gemm(s['act'], *self._wb(fc2), br)
            self._add_branch(br, dp, seq_rows, rows, x_in, out, next_ln)
        return out

    def _mlp_bwd(self, s, dx, x_in, fc1, fc2, rows, ln_name, eps, dp=None, seq_rows=0, next_branch=None):
        """dx (rows, 768) fp32 = gradient of the block output; adds the MLP branch's contribution through LN(x_in) into dx."""
        dy_b = self._branch_dy(dx, dp, seq_rows, rows, fc2 + '.bias')                                     # gradient of the (scaled) branch + fc2's bias gradient
        dact = self._lin_bwd(fc2, dy_b, s['act'], rows, tag='act', bias_done=True, dx_dtype=torch.bfloat16)   # consumed by the bf16 GELU backward only
        dpre = self._buf('dpre', (rows, FF), torch.bfloat16)
        self._gelu_bwd(s['pre'], dact, dpre)
        dh = self._lin_bwd(fc1, dpre, s['h2'], rows, tag='h', dx_dtype=torch.bfloat16)                      # read once, by the LN backward
        self._ln_bwd(x_in, ln_name, dh, dx, rows, eps, next_branch=next_branch, acc_dx=True)

    # ---- divided space-time attention ---------------------------------------------------------------------------------
    @staticmethod
    def _group_maps(kind):
        """Row maps between token order (seg, 1 + f*196 + p) and group-sequence order; logical row r runs over (seg, group, tok)."""
        if kind == 'time':        # groups = spatial positions p, tokens = frames f  ('(b n) f d', vit_helper.py:343-344)
            G, T = 196, 8
            tok = ops.rowmap(VIS_P, T, VIS_L, 1, 196, 1)
        else:                     # groups = frames f, tokens = positions p          ('(b f) n d', vit_helper.py:341-342)
            G, T = 8, 196
            tok = ops.rowmap(VIS_P, T, VIS_L, 196, 1, 1)
        Lg = T + 1
        grp = ops.rowmap(VIS_P, T, G * Lg, Lg, 1, 1)
        cls_tok = ops.rowmap(G, G, VIS_L, 0, 0, 0)
        cls_grp = ops.rowmap(G, G, G * Lg, 0, Lg, 0)
        return G, T, Lg, tok, grp, cls_tok, cls_grp

    def _divided_fwd(self, qkv, att, n, kind, stats=None):
        """stats (optional): fp32 (n * H * 2) - the CLS query's softmax statistics over all keys, kept for sf_attention_{group,tiny}_bwd_clsq."""
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        kw = dict(n_groups=196, row0=1, group_stride=1, tok_stride=196, n_tok=8) if kind == 'time' else \
            dict(n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196)
        if kind == 'space':
            # the CLS query's attention over all 1569 keys (vit_helper.py:126) rides in the space kernel as 8 per-frame softmax partials, merged by a tiny combine
            # launch - as in the inference engine - instead of a second pass over K / V (sf_attention_cls).  Not for the time groups: 196 partial records per
            # (segment, head) make the combine launch slower than the pass it replaces (59 us against 44 us, profiled)
            part = self._buf('cls_part', (n * H * 8 * 66,), torch.float32)
            ops.attention_cls_partial(q, k, v, att, part, n_seq=n, seq_rows=VIS_L, cls_row=0, heads=H, head_dim=HD, scale=0.125, **kw)
            if stats is not None:
                _chk(_lib.load().sf_attention_cls_combine_stats(part.data_ptr(), 8, att.data_ptr(), att.stride(0), VIS_L, 0, n, H, stats.data_ptr(), _st()),
                     'sf_attention_cls_combine_stats')
            else:
                ops.attention_cls_combine(part, att, n_part=8, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=H)
            return
        ops.attention(q, k, v, att, n_seq=n, seq_rows=VIS_L, cls_row=0, heads=H, head_dim=HD, scale=0.125, **kw)
        if stats is not None:
            _chk(_lib.load().sf_attention_cls_stats(q.data_ptr(), VIS_L, 0, k.data_ptr(), v.data_ptr(), qkv.stride(0), VIS_L, 0, VIS_L, att.data_ptr(), att.stride(0),
                                                    VIS_L, 0, n, H, HD, 0.125, stats.data_ptr(), _st()), 'sf_attention_cls_stats')
            return
        ops.attention_cls(q, k, v, att, n_seq=n, q_seq_rows=VIS_L, q_row=0, kv_seq_rows=VIS_L, kv_row0=0, n_keys=VIS_L,
                          out_seq_rows=VIS_L, out_row=0, heads=H, head_dim=HD, scale=0.125)

    def _attn_bwd_chunked(self, Gq, GdO, Gd, nseq, Lg):
        step = max(1, 60000 // H)                                 # batched-GEMM grids take < 65536 (sequence, head) pairs
        for s0 in range(0, nseq, step):
            c = min(step, nseq - s0)
            self.attn_bwd_seq(Gq[s0 * Lg:(s0 + c) * Lg], GdO[s0 * Lg:(s0 + c) * Lg], Gd[s0 * Lg:(s0 + c) * Lg], c, Lg, H, HD)

    def _divided_bwd(self, qkv, dO_b, n, kind, att=None, stats=None):
        """qkv (n*1569, 2304) bf16 saved, dO_b (n*1569, 768) bf16 -> dqkv (n*1569, 2304) bf16.  att / stats (space, optional): the forward's attention output and the
        CLS query's softmax statistics - then the CLS query's backward runs inside the group kernels."""
        G, T, Lg, tok, grp, cls_tok, cls_grp = self._group_maps(kind)
        nseq, rows_g, M = n * G, n * G * Lg, n * VIS_L
        if self.fused_attn_bwd:
            # dedicated kernels: 8 x 9 time groups on VALU, 196 x 197 space groups fused on v_mfma_f32_16x16x16_bf16 (scores recomputed,
            # nothing but dq | dk | dv touches HBM); the CLS key's dk | dv of every group is summed by sf_reduce_groups_bf16
            dqkv = self._buf('dqkv', (M, 3 * D), torch.bfloat16)
            part = self._buf('cls_kv_part', (nseq, 2 * D), torch.bfloat16)
            geo = (196, 1, 1, 196, 8) if kind == 'time' else (8, 1, 196, 1, 196)          # n_groups, row0, group_stride, tok_stride, n_tok
            if stats is not None:
                dqc = self._buf('cls_dq_part', (nseq, D), torch.bfloat16)
                fq = _lib.load().sf_attention_tiny_bwd_clsq if kind == 'time' else _lib.load().sf_attention_group_bwd_clsq
                _chk(fq(qkv.data_ptr(), qkv[:, D:].data_ptr(), qkv[:, 2 * D:].data_ptr(), qkv.stride(0), dO_b.data_ptr(),
                                                             dO_b.stride(0), dqkv.data_ptr(), dqkv[:, D:].data_ptr(), dqkv[:, 2 * D:].data_ptr(), dqkv.stride(0),
                                                             part.data_ptr(), stats.data_ptr(), att.data_ptr(), att.stride(0), dqc.data_ptr(), n, VIS_L, *geo, 0, H, HD,
                                                             0.125, _st()), 'sf_attention_*_bwd_clsq')
                # the CLS row: dk | dv = the sum over the groups' slot-0 rows, dq = the sum of the groups' partial rows
                _chk(_lib.load().sf_reduce_groups_bf16(part.data_ptr(), G * 2 * D, 2 * D, G, dqkv[:, D:].data_ptr(), VIS_L * 3 * D, 2 * D, n, 0, _st()),
                     'sf_reduce_groups_bf16')
                _chk(_lib.load().sf_reduce_groups_bf16(dqc.data_ptr(), G * D, D, G, dqkv.data_ptr(), VIS_L * 3 * D, D, n, 0, _st()), 'sf_reduce_groups_bf16')
                return dqkv
            fn = _lib.load().sf_attention_tiny_bwd if kind == 'time' else _lib.load().sf_attention_group_bwd
            _chk(fn(qkv.data_ptr(), qkv[:, D:].data_ptr(), qkv[:, 2 * D:].data_ptr(), qkv.stride(0), dO_b.data_ptr(), dO_b.stride(0), dqkv.data_ptr(),
                    dqkv[:, D:].data_ptr(), dqkv[:, 2 * D:].data_ptr(), dqkv.stride(0), part.data_ptr(), n, VIS_L, *geo, 0, H, HD, 0.125, _st()),
                 'sf_attention_*_bwd')
            _chk(_lib.load().sf_reduce_groups_bf16(part.data_ptr(), G * 2 * D, 2 * D, G, dqkv[:, D:].data_ptr(), VIS_L * 3 * D, 2 * D, n, 0, _st()),
                 'sf_reduce_groups_bf16')
            self._cls_bwd(qkv, dO_b, dqkv, n, VIS_L, VIS_L, do_seq_rows=VIS_L, accumulate=True)
            return dqkv
        Gq = self._buf('g_qkv', (rows_g, 3 * D), torch.bfloat16)
        copy_rows(qkv, Gq, n * VIS_P, 3 * D, tok, grp)
        copy_rows(qkv, Gq, nseq, 3 * D, cls_tok, cls_grp)
        GdO = self._buf('g_dO', (rows_g, D), torch.bfloat16, zero=True)       # CLS slots keep dO = 0
        copy_rows(dO_b, GdO, n * VIS_P, D, tok, grp)
        Gd = self._buf('g_dqkv', (rows_g, 3 * D), torch.bfloat16)
        self._attn_bwd_chunked(Gq, GdO, Gd, nseq, Lg)
        dqkv = self._buf('dqkv', (M, 3 * D), torch.bfloat16)
        copy_rows(Gd, dqkv, n * VIS_P, 3 * D, grp, tok)
        # the CLS row is a key / value of every group: its dk | dv is the sum over the groups' slot-0 rows
        _chk(_lib.load().sf_reduce_groups_bf16(Gd[:, D:].data_ptr(), G * Lg * 3 * D, Lg * 3 * D, G, dqkv[:, D:].data_ptr(), VIS_L * 3 * D, 2 * D, n, 0,
                                               _st()), 'sf_reduce_groups_bf16')
        # CLS query over all 1569 tokens: dq of the CLS row (=), dk | dv of every row (+=)
        self._cls_bwd(qkv, dO_b, dqkv, n, VIS_L, VIS_L, do_seq_rows=VIS_L, accumulate=True)
        return dqkv

    def _cls_bwd(self, qkv, dO_b, dqkv, n_seq, seq_rows, n_keys, do_seq_rows, accumulate):
        _chk(_lib.load().sf_attention_cls_bwd(qkv.data_ptr(), seq_rows, 0, qkv[:, D:].data_ptr(), qkv[:, 2 * D:].data_ptr(), qkv.stride(0), seq_rows, 0,
                                              n_keys, dO_b.data_ptr(), dO_b.stride(0), do_seq_rows, 0, dqkv.data_ptr(), dqkv[:, D:].data_ptr(),
                                              dqkv[:, 2 * D:].data_ptr(), dqkv.stride(0), n_seq, H, HD, 0.125, int(accumulate), _st()),
             'sf_attention_cls_bwd')

    def _attn_branch_bwd(self, dx, rows, proj, att_saved, qkv_fn, h_saved, x_in, ln_name, eps, qkv_names, dp=None, seq_rows=0, next_branch=None):
        """Common tail of an attention residual branch: dx -> proj backward -> attention backward (qkv_fn) -> qkv linear(s)
        backward -> LN backward accumulated into dx.  `dp`: per-segment stochastic-depth scales of this branch (None = branch always kept)."""
        dy_b = self._branch_dy(dx, dp, seq_rows, rows, proj + '.bias')
        dO_b = self._lin_bwd(proj, dy_b, att_saved, rows, tag='h', bias_done=True, dx_dtype=torch.bfloat16)  # attention output gradient, bf16 for the attention backward
        dqkv = qkv_fn(dO_b)
        if isinstance(qkv_names, str):                                        # one fused (2304, 768) projection
            dh = self._lin_bwd(qkv_names, dqkv, h_saved, rows, tag='h', dx_dtype=torch.bfloat16)
        else:                                                                 # separate query / key / value Linears
            dh = self._buf('dx_qkvsum', (rows, D), torch.float32)
            for j, nm in enumerate(qkv_names):
                self._lin_bwd(nm, dqkv[:, j * D:(j + 1) * D], h_saved, rows, dx_out=dh, acc_dx=j > 0)
        self._ln_bwd(x_in, ln_name, dh, dx, rows, eps, next_branch=next_branch, acc_dx=True)

    # ---- aggregator layer (BaseEncoderLayer over nn.TransformerEncoderLayer, motionformer.py:301-334) -----------------
    def _agg_fwd(self, Z, n_seq, L, p, tag):
        rows = n_seq * L
        s = dict(Z=Z, n_seq=n_seq, L=L)
        s['zn'] = self._buf(f'{tag}_zn', (rows, D), torch.bfloat16)
        ops.layernorm(Z, *self._ln(p + '.norm1'), s['zn'], EPS_VIS)
        s['qkv'] = self._buf(f'{tag}_qkv', (rows, 3 * D), torch.bfloat16)
        ops.gemm(s['zn'], self.b[p + '.self_attn.in_proj_weight'], self.p[p + '.self_attn.in_proj_bias'], s['qkv'])
        q3 = s['qkv']
        s['att'] = self._buf(f'{tag}_att', (n_seq, D), torch.bfloat16)
        ops.attention_cls(q3[:, :D], q3[:, D:2 * D], q3[:, 2 * D:], s['att'], n_seq=n_seq, q_seq_rows=L, q_row=0, kv_seq_rows=L, kv_row0=0,
                          n_keys=L, out_seq_rows=1, out_row=0, heads=H, head_dim=HD, scale=0.125)
        s['y'] = self._buf(f'{tag}_y', (n_seq, D), torch.float32)
        ops.gemm(s['att'], *self._wb(p + '.self_attn.out_proj'), s['y'], residual=Z, r_map=ops.rowmap(1, 1, L, 0, 0, 0))
        out = self._mlp_fwd(s, s['y'], 'yn', p + '.linear1', p + '.linear2', n_seq, p + '.norm2', EPS_VIS, tag)
        return out, s

    def _agg_bwd(self, dout, s, p):
        """dout (n_seq, 768) fp32 -> dZ (n_seq*L, 768) fp32 (gradient w.r.t. [agg_cls; tokens])."""
        n_seq, L = s['n_seq'], s['L']
        rows = n_seq * L
        dy = self._buf('agg_dy', (n_seq, D), torch.float32)
        dy.copy_(dout)
        self._mlp_bwd(s, dy, s['y'], p + '.linear1', p + '.linear2', n_seq, p + '.norm2', EPS_VIS)
        dy_b = self._buf('dy_b', (n_seq, D), torch.bfloat16)
        cast_bf16(dy, dy_b, n_seq, D)
        dO_b = self._lin_bwd(p + '.self_attn.out_proj', dy_b, s['att'], n_seq, tag='h', dy_f32=dy, dx_dtype=torch.bfloat16)
        dqkv = self._buf('dqkv', (rows, 3 * D), torch.bfloat16)
        dqkv.view(n_seq, L, 3 * D)[:, 1:, :D].zero_()                        # dq of rows 1.. is zero (only row 0 queries); dk | dv of every row and dq of row 0 come from _cls_bwd ('=')
        self._cls_bwd(s['qkv'], dO_b, dqkv, n_seq, L, L, do_seq_rows=1, accumulate=False)
        dzn = self._lin_bwd(p + '.self_attn.in_proj', dqkv, s['zn'], rows, tag='h', wkey=p + '.self_attn.in_proj_weight',
                            bkey=p + '.self_attn.in_proj_bias')
        dZ = self._buf('agg_dZ', (rows, D), torch.float32)
        self._ln_bwd(s['Z'], p + '.norm1', dzn, dZ, rows, EPS_VIS)              # writes every row ('=': no zero fill of the buffer first)
        dZ.view(n_seq, L, D)[:, 0].add_(dy)                                  # + the residual path of row 0
        return dZ

    # ---- visual tower -------------------------------------------------------------------------------------------------------
    def _vis_table(self):
        pos, temp = self.p[V + '.pos_embed'][0], self.p[V + '.temp_embed'][0]
        body = (pos[1:].unsqueeze(0) + temp.unsqueeze(1)).reshape(-1, D)       # weight prep, as SynchformerEngine.load_weights
        return torch.cat([pos[:1] + self.p[V + '.cls_token'][0], body], 0).contiguous()

    def _fwd_visual(self, vid):
        """vid (n, 16, 3, 224, 224) u8|f16|bf16|f32 -> aggregator outputs (n*8, 768) fp32 (saved state in self.sv_v)."""
        n = vid.shape[0]
        M = n * VIS_L
        self._clear_pre_ln()                                                 # (no LayerNorm left over from an earlier, interrupted forward)
        sv = self.sv_v = dict(n=n, blocks=[])
        sv['patches'] = self._buf('v_patches', (n * VIS_P, 1536), torch.bfloat16)
        ops.im2col_video(vid.contiguous(), sv['patches'])
        x = self._buf('v_x0', (M, D), torch.float32)
        ops.broadcast_rows(x, self._vis_table(), n_seq=n, dst_seq_rows=VIS_L)
        tokmap = ops.rowmap(VIS_P, VIS_P, VIS_L, 0, 1, 1)
        ops.gemm(sv['patches'], self.b[V + '.patch_embed_3d.proj.weight'].view(D, 1536), self.p[V + '.patch_embed_3d.proj.bias'], x, residual=x,
                 c_map=tokmap, r_map=tokmap)
        for i in range(self.n_vblocks):
            p, t = f'{V}.blocks.{i}', f'v{i}'
            s = dict(x=x)
            for kind, ln, att, key in (('time', 'norm3', 'timeattn', 't'), ('space', 'norm1', 'attn', 's')):
                s['h' + key] = self._ln_into(x, f'{p}.{ln}', f'{t}_h{key}', M, EPS_VIS)
                s['qkv' + key] = self._buf(f'{t}_qkv{key}', (M, 3 * D), torch.bfloat16)
                ops.gemm(s['h' + key], *self._wb(f'{p}.{att}.qkv'), s['qkv' + key])
                s['att' + key] = self._buf(f'{t}_att{key}', (M, D), torch.bfloat16)
                s['cst' + key] = self._buf(f'{t}_cst{key}', (n * H * 2,), torch.float32) if (self.cls_in_group and self.fused_attn_bwd) else None
                self._divided_fwd(s['qkv' + key], s['att' + key], n, kind, stats=s['cst' + key])
                xn = self._buf(f'{t}_x{key}', (M, D), torch.float32)
                dp = s['dp_s'] = self._dp_scales(i, 0, n) if kind == 'space' else None     # time attention has no DropPath (vit_helper.py:367-369)
                if dp is None:
                    ops.gemm(s['att' + key], *self._wb(f'{p}.{att}.proj'), xn, residual=x)
                else:
                    br = self._buf('dp_branch', (M, D), torch.float32)
                    ops.gemm(s['att' + key], *self._wb(f'{p}.{att}.proj'), br)
                    self._add_branch(br, dp, VIS_L, M, x, xn, next_ln=(p + '.norm2', f'{t}_h2', EPS_VIS))
                x = s['x' + key] = xn                                           # xt = after time attention, xs = after space attention
            s['dp_m'] = self._dp_scales(i, 1, n)
            nxt = (f'{V}.blocks.{i + 1}.norm3', f'v{i + 1}_ht', EPS_VIS) if i + 1 < self.n_vblocks else None     # the norm that opens the next block
            x = self._mlp_fwd(s, x, 'h2', p + '.mlp.fc1', p + '.mlp.fc2', M, p + '.norm2', EPS_VIS, t, dp=s['dp_m'], seq_rows=VIS_L, next_ln=nxt)
            sv['blocks'].append(s)
        sv['x_last'] = x
        Z = self._buf('v_Z', (n * 8 * AGG_V, D), torch.float32)
        ops.broadcast_rows(Z, self.p[V + '.spatial_attn_agg.cls_token'].view(1, D), n_seq=n * 8, dst_seq_rows=AGG_V)
        sv['in_map'], sv['z_map'] = ops.rowmap(VIS_P, VIS_P, VIS_L, 0, 1, 1), ops.rowmap(VIS_P, 196, 8 * AGG_V, AGG_V, 1, 1)
        ops.layernorm(x, *self._ln(V + '.norm'), Z, EPS_VIS, rows=n * VIS_P, in_map=sv['in_map'], out_map=sv['z_map'])
        out, sv['agg'] = self._agg_fwd(Z, n * 8, AGG_V, V + '.spatial_attn_agg', 'vagg')
        return out

    def _bwd_visual(self, dout, on_ready=None):
        self._pre_dy.pop(self._ws_prefix, None)                              # (no branch head of an earlier, interrupted backward)
        sv = self.sv_v
        n = sv['n']
        M = n * VIS_L
        agg = V + '.spatial_attn_agg'
        dZ = self._agg_bwd(dout, sv['agg'], agg)
        gz = self._buf('gz', (AGG_V, D), torch.float32)
        self._seqsum(dZ, n * 8, AGG_V, gz)
        self.g[agg + '.cls_token'].view(D).copy_(gz[0])
        dx = self._buf('v_dx', (M, D), torch.float32)
        dx.view(n, VIS_L, D)[:, 0].zero_()                                     # CLS rows are dropped before the final norm: grad 0 (28 rows, not a 135 MB fill; the
                                                                               # row-mapped LayerNorm backward below writes every patch row)
        self._ln_bwd(sv['x_last'], V + '.norm', dZ, dx, n * VIS_P, EPS_VIS, x_map=sv['in_map'], dy_map=sv['z_map'], dx_map=sv['in_map'])
        if on_ready:
            on_ready(self._key_range(V + '.norm.', V + '.spatial_attn_agg.'))
        for i in reversed(range(self.n_vblocks)):
            p, s = f'{V}.blocks.{i}', sv['blocks'][i]
            # every branch's closing LayerNorm backward also writes the head of the branch that follows on dx: space after MLP, time after space, the previous
            # block's MLP after time
            nb_prev = None
            if i > 0:
                pp_, sp_ = f'{V}.blocks.{i - 1}', sv['blocks'][i - 1]
                nb_prev = (sp_['dp_m'], VIS_L, pp_ + '.mlp.fc2.bias')
            self._mlp_bwd(s, dx, s['xs'], p + '.mlp.fc1', p + '.mlp.fc2', M, p + '.norm2', EPS_VIS, dp=s['dp_m'], seq_rows=VIS_L,
                          next_branch=(s['dp_s'], VIS_L, p + '.attn.proj.bias'))
            self._attn_branch_bwd(dx, M, p + '.attn.proj', s['atts'], lambda dO, q=s['qkvs'], a=s['atts'], c=s['csts']: self._divided_bwd(q, dO, n, 'space', att=a, stats=c), s['hs'],
                                  s['xt'], p + '.norm1', EPS_VIS, p + '.attn.qkv', dp=s['dp_s'], seq_rows=VIS_L, next_branch=(None, 0, p + '.timeattn.proj.bias'))
            self._attn_branch_bwd(dx, M, p + '.timeattn.proj', s['attt'], lambda dO, q=s['qkvt'], a=s['attt'], c=s['cstt']: self._divided_bwd(q, dO, n, 'time', att=a, stats=c), s['ht'],
                                  s['x'], p + '.norm3', EPS_VIS, p + '.timeattn.qkv', next_branch=nb_prev)
            if on_ready and i % 3 == 0:                                        # blocks i .. i+2 are final: one ~92 MB bucket
                on_ready(self._key_range(*[f'{V}.blocks.{j}.' for j in range(i, min(i + 3, self.n_vblocks))]))
        # token table: row 0 = cls_token + pos[0]; row 1 + f*196 + p = pos[1 + p] + temp[f]  (video_model_builder.py:248-254)
        gtab = self._buf('gtab', (VIS_L, D), torch.float32)
        self._seqsum(dx, n, VIS_L, gtab)
        self.g[V + '.cls_token'].view(D).copy_(gtab[0])
        gpos = self.g[V + '.pos_embed'][0]
        gpos[0].copy_(gtab[0])
        self._seqsum(gtab[1:], 8, 196, gpos[1:])                                # sum over frames
        ws = self._buf('colsum_ws', (D * 4,), torch.float32)
        for f in range(8):
            colsum(gtab[1 + f * 196: 1 + (f + 1) * 196], 196, D, self.g[V + '.temp_embed'][0, f], ws)
        dtok = self._buf('v_dtok', (n * VIS_P, D), torch.bfloat16)
        ops.gather_rows(dx, dtok, n * VIS_P, in_map=sv['in_map'])
        self._lin_bwd(V + '.patch_embed_3d.proj', dtok, sv['patches'], n * VIS_P, need_dx=False)
        if on_ready:
            on_ready(self._key_range(V + '.cls_token', V + '.pos_embed', V + '.temp_embed', V + '.patch_embed_3d.'))

    # ---- audio tower --------------------------------------------------------------------------------------------------------
    def _aud_table(self, L):
        e = A + '.ast.embeddings'
        tab = self.p[e + '.position_embeddings'][0, :L].clone()
        tab[0] += self.p[e + '.cls_token'][0, 0]
        tab[1] += self.p[e + '.distillation_token'][0, 0]
        return tab.contiguous()

    def _fwd_audio(self, spec):
        """spec (n, F=128, Ta=66) fp32 -> aggregator outputs (n*6, 768) fp32."""
        n, Fa, Ta = spec.shape
        self._clear_pre_ln()
        nf, nt = (Fa - 16) // 10 + 1, (Ta - 16) // 10 + 1
        P, L = nf * nt, nf * nt + 2
        if (P, L, nf + 1) != (AUD_P, AUD_L, AGG_A):
            raise ValueError(f'spectrogram {Fa}x{Ta} gives {L} tokens; this build trains the 128x66 / 74-token configuration')
        M = n * L
        sv = self.sv_a = dict(n=n, layers=[], nt=nt)
        sv['patches'] = self._buf('a_patches', (n * P, 256), torch.bfloat16)
        ops.im2col_spec(spec.contiguous().float(), sv['patches'])
        x = self._buf('a_x0', (M, D), torch.float32)
        ops.broadcast_rows(x, self._aud_table(L), n_seq=n, dst_seq_rows=L)
        sv['tokmap'] = ops.rowmap(P, P, L, 0, 1, 2)
        pe = A + '.ast.embeddings.patch_embeddings.projection'
        ops.gemm(sv['patches'], self.b[pe + '.weight'].view(D, 256), self.p[pe + '.bias'], x, residual=x, c_map=sv['tokmap'], r_map=sv['tokmap'])
        for i in range(self.n_alayers):
            p, t = f'{A}.ast.encoder.layer.{i}', f'a{i}'
            s = dict(x=x)
            s['h1'] = self._buf(f'{t}_h1', (M, D), torch.bfloat16)
            ops.layernorm(x, *self._ln(p + '.layernorm_before'), s['h1'], EPS_AST)
            q3 = s['qkv'] = self._buf(f'{t}_qkv', (M, 3 * D), torch.bfloat16)
            for j, nm in enumerate(('query', 'key', 'value')):
                ops.gemm(s['h1'], *self._wb(f'{p}.attention.attention.{nm}'), q3[:, j * D:(j + 1) * D])
            s['att'] = self._buf(f'{t}_att', (M, D), torch.bfloat16)
            ops.attention(q3[:, :D], q3[:, D:2 * D], q3[:, 2 * D:], s['att'], n_seq=n, seq_rows=L, n_groups=1, row0=0, group_stride=0, tok_stride=1,
                          n_tok=L, cls_row=-1, heads=H, head_dim=HD, scale=0.125)
            s['x2'] = self._buf(f'{t}_x2', (M, D), torch.float32)
            ops.gemm(s['att'], *self._wb(p + '.attention.output.dense'), s['x2'], residual=x)
            x = self._mlp_fwd(s, s['x2'], 'h2', p + '.intermediate.dense', p + '.output.dense', M, p + '.layernorm_after', EPS_AST, t)
            sv['layers'].append(s)
        sv['x_last'] = x
        La = AGG_A
        Z = self._buf('a_Z', (n * nt * La, D), torch.float32)
        ops.broadcast_rows(Z, self.p[A + '.freq_attn_agg.cls_token'].view(1, D), n_seq=n * nt, dst_seq_rows=La)
        sv['z_map'] = ops.rowmap(P, nt, nt * La, 1, La, 1)
        ops.layernorm(x, *self._ln(A + '.ast.layernorm'), Z, EPS_AST, rows=n * P, in_map=sv['tokmap'], out_map=sv['z_map'])
        out, sv['agg'] = self._agg_fwd(Z, n * nt, La, A + '.freq_attn_agg', 'aagg')
        return out

    def _bwd_audio(self, dout):
        self._pre_dy.pop(self._ws_prefix, None)
        sv = self.sv_a
        n, nt = sv['n'], sv['nt']
        L, P = AUD_L, AUD_P
        M = n * L
        agg = A + '.freq_attn_agg'
        dZ = self._agg_bwd(dout, sv['agg'], agg)
        gz = self._buf('gz', (AGG_V, D), torch.float32)
        self._seqsum(dZ, n * nt, AGG_A, gz)
        self.g[agg + '.cls_token'].view(D).copy_(gz[0])
        dx = self._buf('a_dx', (M, D), torch.float32)
        dx.view(n, L, D)[:, :L - P].zero_()                                # the cls / distillation rows take no gradient from the final norm (the patch rows are all written below)
        self._ln_bwd(sv['x_last'], A + '.ast.layernorm', dZ, dx, n * P, EPS_AST, x_map=sv['tokmap'], dy_map=sv['z_map'], dx_map=sv['tokmap'])
        for i in reversed(range(self.n_alayers)):
            p, s = f'{A}.ast.encoder.layer.{i}', sv['layers'][i]
            self._mlp_bwd(s, dx, s['x2'], p + '.intermediate.dense', p + '.output.dense', M, p + '.layernorm_after', EPS_AST,
                          next_branch=(None, 0, p + '.attention.output.dense.bias'))

            def full_bwd(dO, q=s['qkv']):
                dqkv = self._buf('dqkv', (M, 3 * D), torch.bfloat16)
                if self.fused_attn_bwd:                                      # one 74 x 74 "group" per (segment, head), no CLS slot
                    _chk(_lib.load().sf_attention_group_bwd(q.data_ptr(), q[:, D:].data_ptr(), q[:, 2 * D:].data_ptr(), q.stride(0), dO.data_ptr(),
                                                            dO.stride(0), dqkv.data_ptr(), dqkv[:, D:].data_ptr(), dqkv[:, 2 * D:].data_ptr(),
                                                            dqkv.stride(0), None, n, L, 1, 0, 0, 1, L, -1, H, HD, 0.125, _st()),
                         'sf_attention_group_bwd')
                else:
                    self.attn_bwd_seq(q, dO, dqkv, n, L, H, HD)
                return dqkv
            self._attn_branch_bwd(dx, M, p + '.attention.output.dense', s['att'], full_bwd, s['h1'], s['x'], p + '.layernorm_before', EPS_AST,
                                  [f'{p}.attention.attention.{nm}' for nm in ('query', 'key', 'value')],
                                  next_branch=(None, 0, f'{A}.ast.encoder.layer.{i - 1}.output.dense.bias') if i > 0 else None)
        e = A + '.ast.embeddings'
        gtab = self._buf('gtab', (VIS_L, D), torch.float32)[:L]
        self._seqsum(dx, n, L, gtab)
        gpos = self.g[e + '.position_embeddings'][0]
        gpos.zero_()
        gpos[:L].copy_(gtab)
        self.g[e + '.cls_token'].view(D).copy_(gtab[0])
        self.g[e + '.distillation_token'].view(D).copy_(gtab[1])
        dtok = self._buf('a_dtok', (n * P, D), torch.bfloat16)
        ops.gather_rows(dx, dtok, n * P, in_map=sv['tokmap'])
        self._lin_bwd(e + '.patch_embeddings.projection', dtok, sv['patches'], n * P, need_dx=False)

    # ---- contrastive head ------------------------------------------------------------------------------------------------------
    def _pool(self, x, t, n, tag):
        out = self._buf(f'{tag}_feat', (n, D), torch.float32)
        return ops.meanpool_l2norm(x, out, t, True)

    def _pool_bwd(self, x, t, dfeat, n, tag):
        dx = self._buf(f'{tag}_dpool', (n * t, D), torch.float32)
        _chk(_lib.load().sf_meanpool_l2norm768_bwd(x.data_ptr(), x.stride(0), t, dfeat.data_ptr(), dfeat.stride(0), dx.data_ptr(), dx.stride(0), 1, n,
                                                   _st()), 'sf_meanpool_l2norm768_bwd')
        return dx

    def _matmul_nt(self, a, bT_rows, out, scale):
        """out (n, d) = scale * a (n, K) @ b (K, d), given b as its rows (K, d): sf_similarity_f32 wants both operands K-contiguous,
        so the small fp32 operands are re-laid out (transposed, K padded to a multiple of 16) with torch copies - data movement only."""
        n, K = a.shape
        Kp = ((K + 15) // 16) * 16
        ap = self._buf('head_a', (n, Kp), torch.float32, zero=True)
        ap[:, :K].copy_(a)
        bp = self._buf('head_b', (bT_rows.shape[1], Kp), torch.float32, zero=True)
        bp[:, :K].copy_(bT_rows.t())
        return ops.similarity(ap, bp, out, scale)

    def _head(self, vfeat, afeat):
        """AVCLIP.compute_loss (open_clip/model.py:506-525) + its backward -> (dvfeat, dafeat) fp32 (n, 768); fills g[logit_scale]."""
        from .dist import all_gather_pair, reduce_scatter_pair
        n = vfeat.shape[0]
        # the temperature is a by-value launch argument.  Its device -> pinned-host copy was queued BEFORE this step's forward (forward_backward), so waiting for it
        # here waits for the previous step's optimizer only - the launcher stays a whole forward ahead of the GPU instead of draining the queue in mid-step
        if self._ls_pending:
            self._ls_ev.synchronize()
            s, self._ls_pending = float(self._ls_host), False
        else:                                                                   # _head called on its own (tests): read it now
            s = float(self.p['logit_scale'])
        world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        gathered = self.gather_for_loss and world > 1
        v_all, a_all = all_gather_pair(vfeat, afeat) if gathered else (vfeat, afeat)      # one 172 KB message for both modalities
        m = v_all.shape[0]
        sims = [self._buf(f'sim{i}', (n, m), torch.float32) for i in range(2)]
        dsims = [self._buf(f'dsim{i}', (n, m), torch.float32) for i in range(2)]
        ops.similarity(vfeat, a_all, sims[0], 1.0 / s)                          # sim_v2a
        ops.similarity(afeat, v_all, sims[1], 1.0 / s)                          # sim_a2v
        tgt = torch.arange(n, device=self.dev, dtype=torch.int64)              # eye(n, m) targets (open_clip/model.py:512-518)
        losses = self._buf('losses', (2,), torch.float32)
        for i in range(2):
            _chk(_lib.load().sf_cross_entropy(sims[i].data_ptr(), m, tgt.data_ptr(), n, m, losses[i:i + 1].data_ptr(), dsims[i].data_ptr(), m, 0.5,
                                              _st()), 'sf_cross_entropy')
        self.losses = losses
        # d logit_scale = -sum(dsim * sim) / s  (sim = <.,.> / s): host-side scalar reduction of two (n, m) products
        self.g['logit_scale'].copy_(-((dsims[0] * sims[0]).sum() + (dsims[1] * sims[1]).sum()) / s)
        dv = self._buf('dvfeat', (n, D), torch.float32)
        da = self._buf('dafeat', (n, D), torch.float32)
        dv_all = self._buf('dv_all', (m, D), torch.float32)
        da_all = self._buf('da_all', (m, D), torch.float32)
        self._matmul_nt(dsims[0], a_all, dv, 1.0 / s)                           # d sim_v2a / d vfeat
        self._matmul_nt(dsims[1], v_all, da, 1.0 / s)                           # d sim_a2v / d afeat
        self._matmul_nt(dsims[1].t(), afeat, dv_all, 1.0 / s)                   # d sim_a2v / d vfeat_all  (m, D)
        self._matmul_nt(dsims[0].t(), vfeat, da_all, 1.0 / s)                   # d sim_v2a / d afeat_all
        if gathered:                                                            # backward of all_gather = reduce-scatter (sum over ranks), one message
            dv_all, da_all = reduce_scatter_pair(dv_all, da_all, n)
        ops_add = self._buf('head_sum', (2, n, D), torch.float32)
        torch.add(dv, dv_all, out=ops_add[0])                                   # two (n, 768) adds
        torch.add(da, da_all, out=ops_add[1])
        return ops_add[0], ops_add[1]

    # ---- public API ----------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def clamp_logit_scale(self):
        self.p['logit_scale'].clamp_(*self.clamp_scale)                        # open_clip/model.py:569-572

    @torch.no_grad()
    def forward_backward(self, vis: torch.Tensor, aud: torch.Tensor, on_ready=None) -> torch.Tensor:
        """vis (B, S, Tv=16, C=3, H, W) (Synchformer layout) u8|float, aud (B, S, 1, F, Ta) fp32 -> loss (device scalar); fills flat_g."""
        B, S = vis.shape[:2]
        n = B * S
        self.clamp_logit_scale()
        if self._ls_host is None:
            self._ls_host, self._ls_ev = torch.empty((), dtype=torch.float32).pin_memory(), torch.cuda.Event()
        self._ls_host.copy_(self.p['logit_scale'].detach().reshape(()), non_blocking=True)
        self._ls_ev.record()
        self._ls_pending = True
        # (round 5: no flat_g.zero_() - an 857 MB fill per step: every one of the 449 gradients is written with '=' by the backward before anything reads it, proven by
        #  poisoning the buffer with NaN - tests/test_stage1_gpu.py::test_backward_overwrites_every_gradient: 0 of 449 still NaN in both stream modes, with and without
        #  stochastic depth; `_init_flat` packs the parameters back to back, there are no gaps that could keep stale values)
        if os.environ.get('SF_S1_POISON') == '1':
            self.flat_g.fill_(float('nan'))
        self.fwd_count += 1                                                    # a fresh set of stochastic-depth masks per forward pass
        aud3 = aud.reshape(n, aud.shape[-2], aud.shape[-1])
        if not self.two_streams:
            vout = self._fwd_visual(vis.reshape(n, *vis.shape[2:]))
            aout = self._fwd_audio(aud3)
            self.vfeat, self.afeat = self._pool(vout, 8, n, 'v'), self._pool(aout, self.sv_a['nt'], n, 'a')
            dv, da = self._head(self.vfeat, self.afeat)
            self._bwd_audio(self._pool_bwd(aout, self.sv_a['nt'], da, n, 'a'))      # the short tower first: its bucket travels under the long one
            if on_ready:
                on_ready(self._key_range(A + '.', 'logit_scale'))
            self._bwd_visual(self._pool_bwd(vout, 8, dv, n, 'v'), on_ready)
            self.loss = self.losses.mean()
            return self.loss
        # The towers only meet in the contrastive head: the audio tower (2,072 token rows - small, latency-bound launches) runs its forward
        # and its backward on a second stream next to the visual tower, with its own workspaces (`_ws_prefix`).
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
            self._ev = [torch.cuda.Event() for _ in range(4)]
        main, side = torch.cuda.current_stream(), self._side
        self._ev[0].record(main)
        with torch.cuda.stream(side):
            side.wait_event(self._ev[0])
            self._ws_prefix = 'a:'
            aout = self._fwd_audio(aud3)
            self._ws_prefix = ''
            self._ev[1].record(side)
        vout = self._fwd_visual(vis.reshape(n, *vis.shape[2:]))
        main.wait_event(self._ev[1])
        self.vfeat, self.afeat = self._pool(vout, 8, n, 'v'), self._pool(aout, self.sv_a['nt'], n, 'a')
        dv, da = self._head(self.vfeat, self.afeat)
        self._ev[2].record(main)
        with torch.cuda.stream(side):
            side.wait_event(self._ev[2])
            self._ws_prefix = 'a:'
            self._bwd_audio(self._pool_bwd(aout, self.sv_a['nt'], da, n, 'a'))
            self._ws_prefix = ''
            if on_ready:
                on_ready(self._key_range(A + '.', 'logit_scale'))                  # issued from the side stream: the collective waits for THIS tower
            self._ev[3].record(side)
        self._bwd_visual(self._pool_bwd(vout, 8, dv, n, 'v'), on_ready)
        main.wait_event(self._ev[3])
        self.loss = self.losses.mean()
        return self.loss

    def _key_range(self, *prefixes):
        """[lo, hi) of the flat buffers covered by the keys starting with any of `prefixes` (they are contiguous by construction)."""
        if not hasattr(self, '_offsets'):
            self._offsets, o = {}, 0
            for k in self.keys:
                self._offsets[k] = (o, o + self.p[k].numel())
                o += self.p[k].numel()
        spans = [self._offsets[k] for k in self.keys if k.startswith(tuple(prefixes))]
        lo, hi = min(a for a, _ in spans), max(b for _, b in spans)
        assert sum(b - a for a, b in spans) == hi - lo, f'{prefixes}: not contiguous'
        return lo, hi

    def train_step(self, vis: torch.Tensor, aud: torch.Tensor, lr: Optional[float] = None) -> torch.Tensor:
        """One Stage-1 iteration (train_clip_src/training/train.py:103-154): forward, backward, DDP-mean all-reduce, clip + AdamW.
        With a process group the 859 MB of gradients leave in 7 buckets while the backward is still running: the audio
        tower + logit_scale right after its (short) backward, then norm/aggregator, four 3-block groups in reverse order and the token
        tables of the visual tower as they become final; RCCL runs them on its own stream behind an event on the compute stream."""
        from .dist import BucketedAllReduce
        red = BucketedAllReduce(self.flat_g)
        world = red.world
        loss = self.forward_backward(vis, aud, on_ready=lambda span: red.launch(span[0], span[1]))
        if world > 1 and self.time_comm:                                       # exposed communication = how long the compute stream stalls on the buckets
            self._comm_ev = self._comm_ev or (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self._comm_ev[0].record()
        red.wait()
        if world > 1:
            if self.time_comm:
                self._comm_ev[1].record()
            assert red.covered(), 'the gradient buckets do not tile the flat buffer'
        red.finish()                                                           # DDP semantics: mean over ranks
        self.optimizer_step(lr)
        return loss

    def exposed_comm_ms(self) -> float:
        """Milliseconds the compute stream waited for the gradient buckets in the last train_step (0 without a process group)."""
        if self._comm_ev is None:
            return 0.0
        self._comm_ev[1].synchronize()
        return self._comm_ev[0].elapsed_time(self._comm_ev[1])

    def model_state_dict(self) -> Dict[str, torch.Tensor]:
        """Checkpoint in the reference's AVCLIP key names (v_encoder. / a_encoder. / logit_scale)."""
        return {k.replace(V + '.', 'v_encoder.').replace(A + '.', 'a_encoder.'): t.detach().clone() for k, t in self.p.items()}


def shift_and_get_preds(a: torch.Tensor, v: torch.Tensor, W: int):
    """a, v (B, S, D) fp32 segment embeddings on the device -> (preds_a, preds_v) int64 (B, S - W + 1): for every W-segment window
    of A the index of the most similar window of V and vice versa (train_clip_src/training/train.py:549-579).  The window
    similarity is a diagonal sum of the per-clip segment similarity, so one fp32 similarity launch + one tiny kernel do it."""
    assert a.shape == v.shape, f'{a.shape} != {v.shape}'
    B, S, Dm = a.shape
    n = S - W + 1
    G = torch.empty(B * S, B * S, device=a.device, dtype=torch.float32)
    ops.similarity(a.reshape(B * S, Dm).contiguous().float(), v.reshape(B * S, Dm).contiguous().float(), G, 1.0)
    pa = torch.empty(B, n, device=a.device, dtype=torch.int64)
    pv = torch.empty(B, n, device=a.device, dtype=torch.int64)
    _chk(_lib.load().sf_shift_window_preds(G.data_ptr(), G.stride(0), B, S, W, pa.data_ptr(), pv.data_ptr(), _st()), 'sf_shift_window_preds')
    return pa, pv


@torch.no_grad()
def eval_one_example(model, rgb: torch.Tensor, audio: torch.Tensor, win: int = 8):
    """eval_one_example (train_clip_src/training/train.py:405-444) on the drop-in AVCLIP: contrastive loss from
    `forward_for_logging` + the shifted-window zero-shot precision (configs/segment_avclip.yaml: run_shifted_win_val_winsize 8).
    rgb (B, S, C, Tv, H, W), audio (B, S, Ta, F) -> (losses, metrics) dicts of device scalars."""
    assert not model.training, 'Model should be in eval mode.'
    B, S = rgb.shape[:2]
    out = model.forward_for_logging(rgb, audio, for_momentum=False, for_loop=False, do_norm=True)
    assert win < S, f'Win size ({win}) should be < than the number of segments.'
    pa, pv = shift_and_get_preds(out['segment_afeat'].view(B, S, -1), out['segment_vfeat'].view(B, S, -1), win)
    n = pa.shape[-1]
    gt = torch.arange(n, device=pa.device).view(1, n)                       # get_gt (:581-592)
    prec_a, prec_v = (pa == gt).sum(-1) / n, (pv == gt).sum(-1) / n         # calc_cls_metrics (:594-613)
    metrics = {'precision_a': prec_a.mean(), 'precision_v': prec_v.mean(), 'precision': ((prec_a + prec_v) / 2).mean()}
    return {'segment_contrastive_loss': out['segment_contrastive_loss']}, metrics


class AVCLIPTrainFunction(torch.autograd.Function):
    """autograd bridge for the drop-in `AVCLIP` module: the forward runs the whole HIP step (forward + backward, the loss is a
    scalar so its parameter gradients are known up to the incoming scale), the backward hands `grad_output * dloss/dparam` to
    autograd - GradScaler's scaled loss, DistributedDataParallel's reducer hooks and torch.optim.AdamW then work on the
    module's nn.Parameters unchanged (train_clip_src/training/train.py:143-154)."""

    @staticmethod
    def forward(ctx, trainer: AVCLIPTrainer, vis, aud, *params):
        ctx.trainer = trainer
        ctx.fwd_count0 = trainer.fwd_count
        ctx.save_for_backward(vis, aud)
        loss = trainer.forward_backward(vis, aud).clone()
        trainer.generation = ctx.generation = getattr(trainer, 'generation', 0) + 1
        return loss

    @staticmethod
    def backward(ctx, gout):
        tr = ctx.trainer
        if ctx.generation != tr.generation:
            # The gradients live in the trainer's ONE flat buffer and a later grad-enabled forward of the same module has overwritten them (forward, forward,
            # backward, backward).  The step is re-run from the kept inputs under the same stochastic-depth masks (counter restored) - the price of that call
            # pattern is one more forward + backward; the usual forward -> backward order never takes this branch.
            vis, aud = ctx.saved_tensors
            keep = tr.fwd_count
            tr.fwd_count = ctx.fwd_count0
            tr.forward_backward(vis, aud)
            tr.fwd_count = keep
            tr.generation += 1
        return (None, None, None) + tuple(tr.g[k] * gout for k in tr.keys)
