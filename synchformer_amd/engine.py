"""Forward engine: Synchformer.forward() as a fixed schedule of libsynchformer_hip launches.

Host side of the hot path.  It owns (1) weight preparation - bf16 GEMM operands,
fp32 biases/LN parameters, fused q|k|v weights, and the positional tables with their CLS/DISTILL/OFF/MOD rows
folded in - and (2) the launch order and workspace.  Every FLOP and every activation byte is produced by a
kernel of the C ABI (`ops.*`); torch only allocates device memory and provides the stream.

HBM layout (all row-major, 768-wide token matrices; one "sequence" = one 0.64 s segment):
  X   fp32 (n*1569, 768)   residual stream of the visual branch (fp32 like the reference's autocast residual)
  XN  bf16 (n*1576, 768)   LayerNorm output / attention output (GEMM A operands)
  BIG bf16 (n*1576*3072)   qkv (.., 2304) | MLP hidden (.., 3072) | im2col patches (n*1568, 1536), time-shared
  Z   fp32 (n*8*197, 768)  per-frame aggregator sequences [agg_cls; 196 patch tokens]
Segments are processed `seg_chunk` at a time so that X/XN/BIG of a chunk stay L2/Infinity-Cache friendly and the
workspace stays small; chunking changes nothing numerically (segments are independent until vproj/aproj,
the reference's own `for_loop` switch, motionformer.py:200-207).
"""
import math
import os
from typing import Dict, Optional

import torch

from . import ops

D = 768
FF = 3072
VIS_L = 1569          # 1 + 8*196 tokens per visual segment
VIS_P = 1568
AGG_V = 197           # agg cls + 196 patches per frame
AUD_L = 74            # CLS + DISTILL + 12*6 patches
AUD_P = 72
AGG_A = 13            # agg cls + 12 frequency tokens per time step
EPS_VIS, EPS_AST, EPS_SYNC = 1e-6, 1e-12, 1e-5


class _Lin:
    """bf16 weight (N, K) + fp32 bias (N,) on device; `wk` = the same weight k-step-major (K/32, 768, 32) for sf_gemm_res_ln768, made on demand."""
    __slots__ = ('w', 'b', '_wk')

    def __init__(self, w, b, dev):
        self.w = w.detach().to(dev, torch.bfloat16).contiguous()
        self.b = b.detach().to(dev, torch.float32).contiguous() if b is not None else None
        self._wk = None

    @property
    def wk(self):
        if self._wk is None:
            self._wk = ops.kmajor_weight(self.w)
        return self._wk


class _LinQ:
    """MXFP8 weight: e4m3 bytes (N, K) + stage-major E8M0 scale planes (K / 128, N, 4) + fp32 bias, quantised ON the device by sf_quantize_mxfp8."""
    __slots__ = ('q', 's', 'b')

    def __init__(self, lin: '_Lin'):
        N, K = lin.w.shape
        self.q = torch.empty(N, K, device=lin.w.device, dtype=torch.uint8)
        self.s = ops.mx_scale_planes(N, K, lin.w.device)
        ops.quantize_mxfp8(lin.w, self.q, self.s)
        self.b = lin.b


class _LN:
    __slots__ = ('g', 'b')

    def __init__(self, sd, name, dev):
        self.g = sd[name + '.weight'].detach().to(dev, torch.float32).contiguous()
        self.b = sd[name + '.bias'].detach().to(dev, torch.float32).contiguous()


class SynchformerEngine:
    def __init__(self, state_dict: Dict[str, torch.Tensor], device='cuda:0', seg_chunk: int = 224, fp8_towers: bool = False):
        self.dev = torch.device(device)
        if self.dev.type != 'cuda':
            raise RuntimeError('SynchformerEngine needs a HIP device; there is no CPU path in the product')
        self.seg_chunk = seg_chunk
        # fp8_towers: the six big Linears of every visual block run on MXFP8 operands (sf_gemm_mxfp8) - the frozen-extractor mode of the
        # synchronizability fine-tune (BASELINE configs[4]).  Off (bf16) for every other workload: inference parity bars are stated for bf16.
        self.fp8_towers = bool(fp8_towers)
        self.pe_tokens = os.environ.get('SF_PE_TOKENS', '1') != '0'      # patch embedding on the token layout (identity row maps, persistent GEMM)
        self.fuse_mx_time = True                     # fp8 towers: sf_qkv_time_attention_mx instead of sf_gemm_mxfp8 + the time attention kernels
        self.fuse_mx_attn = os.environ.get('SF_MX_ATTN', '1') != '0'   # fp8 towers: the space attention writes MXFP8 itself (sf_attention_cls_partial_mx; off = bf16 output + sf_quantize_mxfp8)
        self.fuse_mx_ln = True                       # fp8 towers: sf_gemm_mx_res_ln768 instead of sf_gemm_mxfp8 + sf_layernorm768_mxfp8 (tests switch it off to compare)
        # fp8 towers, mixed policy (measurement: DESIGN 4 / profiles/r06_mxfp8_policy.md): the Linears named here keep bf16 operands - 'proj' (both attention projections)
        # and / or 'fc2'; their outputs are re-quantised by sf_quantize_mxfp8 for the next MX launch.  Empty = every big Linear on MXFP8 (the product's fp8_towers mode).
        self.mx_bf16 = frozenset(x for x in os.environ.get('SF_MX_BF16', '').split(',') if x)
        self.capture_blocks = None          # tests: a dict -> the fp32 residual stream after each visual block is cloned into it (key = block index)
        self._ws = {}
        self.audio_side_stream = os.environ.get('SF_AUDIO_SIDE_STREAM', '1') != '0'
        self.fuse_ln = os.environ.get('SF_FUSE_LN', '1') != '0'            # A/B switches of the full-row GEMM + residual + LayerNorm kernel
        self.fuse_ln_fc2 = os.environ.get('SF_FUSE_LN_FC2', '1') != '0'   # K = 3072: 1932 us fused vs 1708 + 277 us (profiles/r02_gemm_ln.md)
        self.fuse_time = os.environ.get('SF_FUSE_TIME', '1') != '0'       # temporal qkv projection + time attention in one launch (sf_qkv_time_attention)
        self.fuse_space = os.environ.get('SF_FUSE_SPACE', '1') != '0'     # spatial qkv projection + space attention in one launch (sf_qkv_space_attention, round 4)
        self.fuse_time2 = os.environ.get('SF_FUSE_TIME2', '1') != '0'     # the temporal launch on the spatial kernel's 192 x 384 main loop (sf_qkv_time_attention2, round 4)
        self._a_side = None
        # Small batches: the launches of a block leave a fifth of the CU-time idle (one clip = 14 segments, M = 21,966 rows: 172 full-row tiles of sf_gemm_res_ln768
        # for 256 CUs, 4.03 / 2.95 / 2.6 rounds in fc1 / the two attention launches), and a tile cannot be made smaller without streaming W again
        # (profiles/r06_small_m.md).  Between `vis_split_min` and `vis_split_max` segments (one to eight clips) the visual tower runs as TWO independent halves of the segments
        # on two HIP streams (segments are independent until vproj, motionformer.py:200-207), one half's partial rounds beside the other's: under a HIP graph -7 % at two clips,
        # -5 % at three, -1..2 % at four to eight, +0.3 % at sixteen, -1.2 .. -2.5 % at one clip on three boxes (tools/r06_split_window.py): hence the window.
        self.vis_split_min = int(os.environ.get('SF_VIS_SPLIT_MIN', '14'))
        self.vis_split_max = int(os.environ.get('SF_VIS_SPLIT_MAX', '112'))
        # 'graph' (default): only inside capture() - a replayed HIP graph has no per-launch host cost, there the split is a clean -4 %; issued eagerly the two halves are twice
        # the launches for the host, and whether the GPU-side gain survives that depends on the box's CPU (measured -2 % on one box, +8 % on another).  'always' / 'never' override.
        self.vis_split_mode = os.environ.get('SF_VIS_SPLIT', 'graph')
        self._in_capture = False
        self.vis_split_parts = int(os.environ.get('SF_VIS_SPLIT_PARTS', '2'))
        self._ws_tag = ''
        self._v_side = None
        self.load_weights(state_dict)

    # ------------------------------------------------------------------------------------------------
    # weight preparation (one-off; not on the hot path)
    # ------------------------------------------------------------------------------------------------
    def load_weights(self, sd: Dict[str, torch.Tensor]):
        self.load_tower_weights(sd)
        self.load_sync_weights(sd)

    def load_tower_weights(self, sd: Dict[str, torch.Tensor]):
        """The two feature extractors (214.8M parameters; frozen in Stage-2 training, train_utils.py:199-204)."""
        dev = self.dev
        f32 = lambda k: sd[k].detach().to(dev, torch.float32)
        lin = lambda k: _Lin(sd[k + '.weight'], sd[k + '.bias'], dev)
        v = 'vfeat_extractor'
        self.v_pe = _Lin(sd[f'{v}.patch_embed_3d.proj.weight'].reshape(D, -1), sd[f'{v}.patch_embed_3d.proj.bias'], dev)
        # sign of filter 0 per patch element: the reference's inf -> NaN token-mask rule (video_model_builder.py:185-201)
        self.v_w0_sign = torch.sign(f32(f'{v}.patch_embed_3d.proj.weight')[0].reshape(-1)).to(torch.int8).contiguous()
        pos, temp = f32(f'{v}.pos_embed')[0], f32(f'{v}.temp_embed')[0]
        body = (pos[1:].unsqueeze(0) + temp.unsqueeze(1)).reshape(-1, D)       # row f*196+n (vmb:248-254)
        self.v_table = torch.cat([pos[:1] + f32(f'{v}.cls_token')[0], body], 0).contiguous()   # (1569, 768)
        self.v_table_pe = self.v_table.clone()                                                   # ... for the token-layout patch embedding (extract_vfeats): its
        self.v_table_pe[0] -= self.v_pe.b                                                        # GEMM adds the bias to the CLS rows as well
        self.v_blocks = []
        i = 0
        while f'{v}.blocks.{i}.norm1.weight' in sd:
            b = f'{v}.blocks.{i}'
            self.v_blocks.append(dict(
                norm1=_LN(sd, b + '.norm1', dev), norm2=_LN(sd, b + '.norm2', dev), norm3=_LN(sd, b + '.norm3', dev),
                t_qkv=lin(b + '.timeattn.qkv'), t_proj=lin(b + '.timeattn.proj'),
                s_qkv=lin(b + '.attn.qkv'), s_proj=lin(b + '.attn.proj'),
                fc1=lin(b + '.mlp.fc1'), fc2=lin(b + '.mlp.fc2')))
            if self.fp8_towers:
                blk = self.v_blocks[-1]
                blk['mx'] = {k: _LinQ(blk[k]) for k in ('t_qkv', 't_proj', 's_qkv', 's_proj', 'fc1', 'fc2')}
            i += 1
        self.v_norm = _LN(sd, f'{v}.norm', dev)
        self.v_agg = self._agg(sd, f'{v}.spatial_attn_agg')
        a = 'afeat_extractor'
        e = f'{a}.ast.embeddings'
        self.a_pe = _Lin(sd[f'{e}.patch_embeddings.projection.weight'].reshape(D, -1),
                         sd[f'{e}.patch_embeddings.projection.bias'], dev)
        self.a_w0_sign = torch.sign(f32(f'{e}.patch_embeddings.projection.weight')[0].reshape(-1)).to(torch.int8).contiguous()
        tab = f32(f'{e}.position_embeddings')[0].clone()
        tab[0] += f32(f'{e}.cls_token')[0, 0]
        tab[1] += f32(f'{e}.distillation_token')[0, 0]
        self.a_table = tab.contiguous()                                          # (74, 768)
        self.a_layers = []
        i = 0
        while f'{a}.ast.encoder.layer.{i}.layernorm_before.weight' in sd:
            L = f'{a}.ast.encoder.layer.{i}'
            att = L + '.attention.attention'
            qkv_w = torch.cat([sd[f'{att}.{n}.weight'] for n in ('query', 'key', 'value')], 0)
            qkv_b = torch.cat([sd[f'{att}.{n}.bias'] for n in ('query', 'key', 'value')], 0)
            self.a_layers.append(dict(
                ln1=_LN(sd, L + '.layernorm_before', dev), ln2=_LN(sd, L + '.layernorm_after', dev),
                qkv=_Lin(qkv_w, qkv_b, dev), o=lin(L + '.attention.output.dense'),
                fc1=lin(L + '.intermediate.dense'), fc2=lin(L + '.output.dense')))
            i += 1
        self.a_norm = _LN(sd, f'{a}.ast.layernorm', dev)
        self.a_agg = self._agg(sd, f'{a}.freq_attn_agg')

    def load_sync_weights(self, sd: Dict[str, torch.Tensor]):
        """vproj / aproj / sync transformer (22.6M parameters: the part Stage-2 trains).  Cheap enough to refresh after every
        optimizer step of an external optimizer without touching the tower operands or the workspaces."""
        dev = self.dev
        # new operand tensors replace the old ones (their allocator blocks are released): a HIP graph captured before this call still points at the OLD
        # blocks, so captured replays are invalidated through this counter (capture() checks it) instead of silently reading freed memory
        self._sync_generation = getattr(self, '_sync_generation', 0) + 1
        f32 = lambda k: sd[k].detach().to(dev, torch.float32)
        lin = lambda k: _Lin(sd[k + '.weight'], sd[k + '.bias'], dev)
        self.vproj, self.aproj = lin('vproj'), lin('aproj')
        t = 'transformer'
        self.s_vln, self.s_aln = _LN(sd, f'{t}.vis_in_lnorm', dev), _LN(sd, f'{t}.aud_in_lnorm', dev)
        self.s_pos = f32(f'{t}.pos_emb_cfg.pos_emb')[0].contiguous()
        self.s_off, self.s_mod = f32(f'{t}.OFF_tok')[0, 0], f32(f'{t}.MOD_tok')[0, 0]
        self._s_tables = {}
        self.s_blocks = []
        i = 0
        while f'{t}.blocks.{i}.ln1.weight' in sd:
            b = f'{t}.blocks.{i}'
            qkv_w = torch.cat([sd[f'{b}.attn.{n}.weight'] for n in ('query', 'key', 'value')], 0)
            qkv_b = torch.cat([sd[f'{b}.attn.{n}.bias'] for n in ('query', 'key', 'value')], 0)
            self.s_blocks.append(dict(ln1=_LN(sd, b + '.ln1', dev), ln2=_LN(sd, b + '.ln2', dev),
                                      qkv=_Lin(qkv_w, qkv_b, dev), proj=lin(b + '.attn.proj'),
                                      fc1=lin(b + '.mlp.0'), fc2=lin(b + '.mlp.2')))
            i += 1
        self.s_heads = 8
        self.s_lnf = _LN(sd, f'{t}.ln_f', dev)
        head = 'off_head' if f'{t}.off_head.weight' in sd else 'sync_head'
        self.s_head = lin(f'{t}.{head}')
        self.n_out = self.s_head.w.shape[0]

    def _agg(self, sd, p):
        dev = self.dev
        return dict(cls=sd[p + '.cls_token'].detach().to(dev, torch.float32).reshape(1, D).contiguous(),
                    norm1=_LN(sd, p + '.norm1', dev), norm2=_LN(sd, p + '.norm2', dev),
                    qkv=_Lin(sd[p + '.self_attn.in_proj_weight'], sd[p + '.self_attn.in_proj_bias'], dev),
                    o=_Lin(sd[p + '.self_attn.out_proj.weight'], sd[p + '.self_attn.out_proj.bias'], dev),
                    fc1=_Lin(sd[p + '.linear1.weight'], sd[p + '.linear1.bias'], dev),
                    fc2=_Lin(sd[p + '.linear2.weight'], sd[p + '.linear2.bias'], dev))

    def _sync_table(self, Sv: int, Sa: int):
        key = (Sv, Sa)
        if key not in self._s_tables:
            L = 2 + Sv + Sa
            if L > self.s_pos.shape[0]:
                raise ValueError(f'sequence of {L} tokens exceeds pos_emb length {self.s_pos.shape[0]}')
            tab = self.s_pos[:L].clone()
            tab[0] += self.s_off
            tab[1 + Sv] += self.s_mod
            self._s_tables[key] = tab.contiguous()
        return self._s_tables[key]

    # ------------------------------------------------------------------------------------------------
    # workspace
    # ------------------------------------------------------------------------------------------------
    def _buf(self, name, numel, dtype):
        name = self._ws_tag + name                                     # (a half of a small batch running on the second visual stream has workspaces of its own)
        t = self._ws.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.zeros(numel, device=self.dev, dtype=dtype)
            self._ws[name] = t
        return t[:numel]

    # ------------------------------------------------------------------------------------------------
    # shared sub-schedules
    # ------------------------------------------------------------------------------------------------
    def _agg_layer(self, Z, n_seq, L, agg, out, tag, key_keep=None):
        """BaseEncoderLayer (motionformer.py:301-334): Z fp32 (n_seq*L, 768) already holds [agg_cls; tokens].
        Only output row 0 of each sequence is ever read (:332), so everything after K/V is computed for row 0 only."""
        rows = n_seq * L
        sfx = 'a' if tag == 'aagg' else ''                              # audio-side workspaces, see extract_afeats
        zn = self._buf('XN' + sfx, rows * D, torch.bfloat16).view(rows, D)
        qkv = self._buf('BIG' + sfx, rows * 3 * D, torch.bfloat16).view(rows, 3 * D)
        ops.layernorm(Z, agg['norm1'].g, agg['norm1'].b, zn, EPS_VIS)
        ops.gemm(zn, agg['qkv'].w, agg['qkv'].b, qkv)
        att = self._buf(tag + '_att', n_seq * D, torch.bfloat16).view(n_seq, D)
        ops.attention_cls(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], att, n_seq=n_seq, q_seq_rows=L, q_row=0,
                          kv_seq_rows=L, kv_row0=0, n_keys=L, out_seq_rows=1, out_row=0, heads=12, head_dim=64,
                          scale=0.125, key_keep=key_keep)
        y = self._buf(tag + '_y', n_seq * D, torch.float32).view(n_seq, D)
        ops.gemm(att, agg['o'].w, agg['o'].b, y, residual=Z, r_map=ops.rowmap(1, 1, L, 0, 0, 0))
        yn = self._buf(tag + '_yn', n_seq * D, torch.bfloat16).view(n_seq, D)
        ops.layernorm(y, agg['norm2'].g, agg['norm2'].b, yn, EPS_VIS)
        h = self._buf(tag + '_h', n_seq * FF, torch.bfloat16).view(n_seq, FF)
        ops.gemm(yn, agg['fc1'].w, agg['fc1'].b, h, gelu=True)
        ops.gemm(h, agg['fc2'].w, agg['fc2'].b, out, residual=y)
        return out

    def _encoder_layer(self, X, rows, xn, big, ln1, qkv, attn_fn, proj, ln2, fc1, fc2, eps):
        """pre-LN transformer layer on the fp32 residual stream X (rows, 768): AST layer / sync Block."""
        ops.layernorm(X, ln1.g, ln1.b, xn, eps, rows=rows)
        q3 = big[:rows * 3 * D].view(rows, 3 * D)
        ops.gemm(xn, qkv.w, qkv.b, q3, M=rows)
        attn_fn(q3, xn)
        ops.gemm(xn, proj.w, proj.b, X, M=rows, residual=X)
        ops.layernorm(X, ln2.g, ln2.b, xn, eps, rows=rows)
        h = big[:rows * FF].view(rows, FF)
        ops.gemm(xn, fc1.w, fc1.b, h, M=rows, gelu=True)
        ops.gemm(h, fc2.w, fc2.b, X, M=rows, residual=X)

    # ------------------------------------------------------------------------------------------------
    # visual branch
    # ------------------------------------------------------------------------------------------------
    def _visual_chunk(self, vid, out, clip_seg=None, keep=None):
        """vid (n, 16, 3, 224, 224) u8|f16|bf16|f32 on device -> out fp32 (n*8, 768).  a3-a9 of SURVEY §8a.
        With clip_seg = (frame0, seg_stride, n_seg), vid is (clips, T, 3, 224, 224) and the segments are read in place."""
        n = vid.shape[0] if clip_seg is None else vid.shape[0] * clip_seg[2]
        rows = n * VIS_L
        X = self._buf('X', rows * D, torch.float32).view(rows, D)
        xn = self._buf('XN', n * 8 * AGG_V * D, torch.bfloat16)[:rows * D].view(rows, D)
        big = self._buf('BIG', n * 8 * AGG_V * FF, torch.bfloat16)
        if rows >= 128 * 64 and self.pe_tokens:
            # patches in the TOKEN layout (a zero row in every segment's CLS slot): the patch-embedding GEMM runs with identity row maps on the persistent kernel
            # (with the row maps it took the 128 x 128 kernel: 1.5 ms per 224-segment launch at 0.55 PFLOP/s).  The CLS rows come out as table + 0 W + bias; the
            # table copy used here has the bias taken off its CLS entry
            patches = big[:rows * 1536].view(rows, 1536)
            if clip_seg is None:
                ops.im2col_video_tokens(vid, patches)
            else:
                ops.im2col_video_tokens(vid, patches, *clip_seg)
            ops.broadcast_rows(X, self.v_table_pe, n_seq=n, dst_seq_rows=VIS_L)
            ops.gemm(patches, self.v_pe.w, self.v_pe.b, X, residual=X)
        else:
            patches = big[:n * VIS_P * 1536].view(n * VIS_P, 1536)
            if clip_seg is None:
                ops.im2col_video(vid, patches)
            else:
                ops.im2col_video_clips(vid, patches, *clip_seg)
            ops.broadcast_rows(X, self.v_table, n_seq=n, dst_seq_rows=VIS_L)
            tokmap = ops.rowmap(VIS_P, VIS_P, VIS_L, 0, 1, 1)
            ops.gemm(patches, self.v_pe.w, self.v_pe.b, X, residual=X, c_map=tokmap, r_map=tokmap)
        qkv = big[:rows * 3 * D].view(rows, 3 * D)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        hid = big[:rows * FF].view(rows, FF)

        part = self._buf('cls_part', n * 12 * 196 * 66, torch.float32)

        fuse_mode = os.environ.get('SF_CLS_FUSION', 'space')          # A/B switch: both | space | none (profiles/r01_notes.md)
        tok_keep = None
        if keep is not None:                                            # content mask (n, 16, 3, 224, 224) bool -> token keep flags
            tok_keep = ops.token_mask_video(keep, self.v_w0_sign, torch.empty(rows, device=self.dev, dtype=torch.uint8))
            if fuse_mode == 'both':                                     # the tiny-group kernel's partial mode has no masked entry point: time goes un-fused
                fuse_mode = 'space'                                     # (large batches fuse it into its projection anyway, below)

        def divided(kind):
            if fuse_mode == 'none' or (fuse_mode == 'space' and kind == 'time'):
                kw = dict(n_groups=196, row0=1, group_stride=1, tok_stride=196, n_tok=8) if kind == 'time' else \
                    dict(n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196)
                ops.attention(q, k, v, xn, n_seq=n, seq_rows=VIS_L, cls_row=0, heads=12, head_dim=64, scale=0.125, key_keep=tok_keep, **kw)
                ops.attention_cls(q, k, v, xn, n_seq=n, q_seq_rows=VIS_L, q_row=0, kv_seq_rows=VIS_L, kv_row0=0, n_keys=VIS_L,
                                  out_seq_rows=VIS_L, out_row=0, heads=12, head_dim=64, scale=0.125, key_keep=tok_keep)
                return
            # patches attend [CLS; their group]; the CLS query's attention over ALL tokens (vit_helper.py:126) is accumulated as
            # per-group partials inside the same kernels (no second pass over K/V) and merged by a tiny combine kernel
            if kind == 'time':   # '(b n) f d' groups (vit_helper.py:343-344)
                ops.attention_cls_partial(q, k, v, xn, part, n_seq=n, seq_rows=VIS_L, n_groups=196, row0=1, group_stride=1,
                                          tok_stride=196, n_tok=8, cls_row=0, heads=12, head_dim=64, scale=0.125)
                groups = 196
            else:                # '(b f) n d' groups (vit_helper.py:341-342)
                ops.attention_cls_partial(q, k, v, xn, part, n_seq=n, seq_rows=VIS_L, n_groups=8, row0=1, group_stride=196,
                                          tok_stride=1, n_tok=196, cls_row=0, heads=12, head_dim=64, scale=0.125, key_keep=tok_keep)
                groups = 8
            ops.attention_cls_combine(part, xn, n_part=groups, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)

        if self.fp8_towers:
            self._visual_blocks_mxfp8(X, xn, qkv, rows, divided, n, part, tok_keep)
            return self._visual_tail(X, n, out, tok_keep)
        # DividedSpaceTimeBlock.forward (vit_helper.py:364-376).  With `fuse_ln` every residual GEMM also emits the LayerNorm that opens the next
        # sub-layer (sf_gemm_res_ln768: the fp32 stream is read and written once per sub-layer, no separate LayerNorm launch); the last block's
        # fc2 stays un-fused because the norm after it is the row-mapped final norm below.
        fuse_ln = self.fuse_ln and rows >= 128 * 64
        fuse_time = self.fuse_time and rows >= 128 * 64
        # (round 5: token-masked forwards run the same fused launches - sf_qkv_space_attention_masked / sf_qkv_time_attention2_masked take the key flags)
        fuse_space = self.fuse_space and rows >= 128 * 64 and fuse_mode != 'none'
        fuse_time2 = fuse_time and self.fuse_time2 and fuse_mode != 'none'
        if fuse_space or fuse_time2:
            side_in = self._buf('side_in', n * 33 * D, torch.bfloat16).view(n * 33, D)
            side = self._buf('side', n * 33 * 3 * D, torch.bfloat16).view(n * 33, 3 * D)
        att = big[:rows * D].view(rows, D)                                        # the time block's attention output (its qkv never exists)
        qkv_cls = self._buf('qkv_cls', n * 3 * D, torch.bfloat16).view(n, 3 * D)
        nb = len(self.v_blocks)
        for bi, b in enumerate(self.v_blocks):
            if bi == 0 or not fuse_ln:
                ops.layernorm(X, b['norm3'].g, b['norm3'].b, xn, EPS_VIS)
            if fuse_time2:
                # temporal qkv + time attention in one launch on 24-patch blocks (sf_qkv_time_attention2); the rows it does not project itself - the CLS row and
                # patches 192..195 of every frame (196 = 8 x 24 + 4): the same 33 rows per segment as in the spatial half - go through a small GEMM up front.
                ops.space_side_rows(xn, side_in, n)
                ops.gemm(side_in, b['t_qkv'].w, b['t_qkv'].b, side)
                ops.qkv_time_attention2(xn, b['t_qkv'].w, b['t_qkv'].b, side, att, part, n_seq=n, scale=0.125, key_keep=tok_keep)
                ops.attention_cls_combine(part, att, n_part=33, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
                t_out = att
            elif fuse_time:
                # temporal qkv + time attention in one launch (sf_qkv_time_attention): the 2304-wide projection never reaches HBM.  The CLS rows'
                # own projection (their k / v are every patch's first key, their q is the global CLS query) is a 224-row GEMM up front.
                ops.gemm(xn.view(n, VIS_L, D)[:, 0], b['t_qkv'].w, b['t_qkv'].b, qkv_cls)
                ops.qkv_time_attention(xn, b['t_qkv'].w, b['t_qkv'].b, qkv_cls, att, part, n_seq=n, n_groups=196, scale=0.125, key_keep=tok_keep)
                ops.attention_cls_combine(part, att, n_part=49, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
                t_out = att
            else:
                ops.gemm(xn, b['t_qkv'].w, b['t_qkv'].b, qkv)
                divided('time')
                t_out = xn
            if fuse_ln:
                ops.gemm_res_ln(t_out, b['t_proj'].wk, b['t_proj'].b, X, b['norm1'].g, b['norm1'].b, xn, EPS_VIS)
            else:
                ops.gemm(t_out, b['t_proj'].w, b['t_proj'].b, X, residual=X)
                ops.layernorm(X, b['norm1'].g, b['norm1'].b, xn, EPS_VIS)
            if fuse_space:
                # spatial qkv + space attention in one launch (sf_qkv_space_attention): the 2304-wide projection never reaches HBM.  The rows the launch does not
                # project itself - the CLS row and the last 4 tokens of every frame (196 = 6 x 32 + 4) - go through a 33-rows-per-segment GEMM up front.
                ops.space_side_rows(xn, side_in, n)
                ops.gemm(side_in, b['s_qkv'].w, b['s_qkv'].b, side)
                ops.qkv_space_attention(xn, b['s_qkv'].w, b['s_qkv'].b, side, att, part, n_seq=n, scale=0.125, key_keep=tok_keep)
                ops.attention_cls_combine(part, att, n_part=8, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
                s_out = att
            else:
                ops.gemm(xn, b['s_qkv'].w, b['s_qkv'].b, qkv)
                divided('space')
                s_out = xn
            if fuse_ln:
                ops.gemm_res_ln(s_out, b['s_proj'].wk, b['s_proj'].b, X, b['norm2'].g, b['norm2'].b, xn, EPS_VIS)
            else:
                ops.gemm(s_out, b['s_proj'].w, b['s_proj'].b, X, residual=X)
                ops.layernorm(X, b['norm2'].g, b['norm2'].b, xn, EPS_VIS)
            ops.gemm(xn, b['fc1'].w, b['fc1'].b, hid, gelu=True)
            if fuse_ln and self.fuse_ln_fc2 and bi + 1 < nb:
                nx = self.v_blocks[bi + 1]['norm3']
                ops.gemm_res_ln(hid, b['fc2'].wk, b['fc2'].b, X, nx.g, nx.b, xn, EPS_VIS)
            else:
                ops.gemm(hid, b['fc2'].w, b['fc2'].b, X, residual=X)
                if fuse_ln and bi + 1 < nb:
                    nx = self.v_blocks[bi + 1]['norm3']
                    ops.layernorm(X, nx.g, nx.b, xn, EPS_VIS)
            if self.capture_blocks is not None:
                self.capture_blocks[bi] = X.clone()
        self._visual_tail(X, n, out, tok_keep)

    def _visual_blocks_mxfp8(self, X, xn, qkv, rows, divided, n, part, tok_keep):
        """The 12 DividedSpaceTimeBlocks with every big Linear on MXFP8 operands.  Activations are quantised where they are produced when the
        producer is ours to change (LayerNorm -> sf_layernorm768_mxfp8, fc1 + GELU -> the GEMM's own MXFP8 epilogue); the attention kernels
        write bf16, which one sf_quantize_mxfp8 pass converts.  The residual stream, LayerNorm statistics and attention stay as in the bf16 path."""
        xq = self._buf('XQ', rows * D, torch.uint8).view(rows, D)
        rows_p = ((rows + 255) // 256) * 256                               # scale planes are padded to whole 256-row tiles (ops.mx_scale_planes)
        xs = self._buf('XS', 6 * rows_p * 4, torch.uint8).view(6, rows_p, 4)
        hq = self._buf('HQ', rows * FF, torch.uint8).view(rows, FF)
        hs = self._buf('HS', 24 * rows_p * 4, torch.uint8).view(24, rows_p, 4)
        fuse = self.fuse_mx_ln                                            # proj / fc2 + residual + the next LayerNorm + its quantisation in one launch (sf_gemm_mx_res_ln768)
        # temporal qkv + time attention + CLS partials in one launch (sf_qkv_time_attention_mx); the CLS rows' own projection is a small MX GEMM on gathered
        # copies of those rows and of their scale dwords
        fuse_time = self.fuse_time and self.fuse_mx_time and rows >= 128 * 64 and tok_keep is None
        if fuse_time:
            n_p = ((n + 255) // 256) * 256
            cq = self._buf('CQ', n * D, torch.uint8).view(n, D)
            cs = self._buf('CS', 6 * n_p * 4, torch.uint8).view(6, n_p, 4)
            qkv_cls = self._buf('qkv_cls', n * 3 * D, torch.bfloat16).view(n, 3 * D)
            aq = self._buf('AQ', rows * D, torch.uint8).view(rows, D)         # the time attention's output as the projection's MXFP8 operand
            as_ = self._buf('AS', 6 * rows_p * 4, torch.uint8).view(6, rows_p, 4)
        fuse_attn = self.fuse_mx_attn and tok_keep is None and os.environ.get('SF_CLS_FUSION', 'space') != 'none'
        # round 4: spatial qkv + space attention in one launch on the MX operands as well (sf_qkv_space_attention_mx, MXFP8 output); round 5: the temporal half on the same
        # schedule (sf_qkv_time_attention2_mx).  The side rows - CLS + the last 4 tokens of every frame - go through sf_gemm_mxfp8 on copies of the rows AND of their scale
        # dwords gathered by one launch (sf_side_rows)
        fuse_space = self.fuse_space and fuse_attn and rows >= 128 * 64
        fuse_time2 = fuse_time and fuse_attn and self.fuse_time2 and rows >= 128 * 64
        if fuse_space or fuse_time2:
            if not fuse_time:
                aq = self._buf('AQ', rows * D, torch.uint8).view(rows, D)
                as_ = self._buf('AS', 6 * rows_p * 4, torch.uint8).view(6, rows_p, 4)
            n33 = n * 33
            n33p = ((n33 + 255) // 256) * 256
            sq = self._buf('SQ', n33 * D, torch.uint8).view(n33, D)
            ss = self._buf('SS', 6 * n33p * 4, torch.uint8).view(6, n33p, 4)
            side = self._buf('side', n33 * 3 * D, torch.bfloat16).view(n33, 3 * D)
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        nb = len(self.v_blocks)
        # mixed policy (self.mx_bf16): bf16 views of BIG for the attention output / the MLP hidden, as in the bf16 schedule
        proj_bf16 = 'proj' in self.mx_bf16 and fuse and fuse_space and fuse_time2
        fc2_bf16 = 'fc2' in self.mx_bf16 and fuse and rows >= 128 * 64
        if proj_bf16 or fc2_bf16:
            big = self._buf('BIG', n * 8 * AGG_V * FF, torch.bfloat16)
            att_b, hid_b = big[:rows * D].view(rows, D), big[:rows * FF].view(rows, FF)
        for bi, b in enumerate(self.v_blocks):
            mx = b['mx']
            if bi == 0 or not fuse:
                ops.layernorm_mxfp8(X, b['norm3'].g, b['norm3'].b, xq, xs, EPS_VIS)
            if fuse_time2 and proj_bf16:
                ops.space_side_rows_mx(xq, xs, sq, ss, n)
                ops.gemm_mxfp8(sq, ss, mx['t_qkv'].q, mx['t_qkv'].s, mx['t_qkv'].b, side)
                ops.qkv_time_attention2_mx(xq, xs, mx['t_qkv'].q, mx['t_qkv'].s, mx['t_qkv'].b, side, att_b, part, n_seq=n, scale=0.125)
                ops.attention_cls_combine(part, att_b, n_part=33, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
                ops.gemm_res_ln(att_b, b['t_proj'].wk, b['t_proj'].b, X, b['norm1'].g, b['norm1'].b, xn, EPS_VIS)
                ops.quantize_mxfp8(xn, xq, xs)
            elif fuse_time2:
                ops.space_side_rows_mx(xq, xs, sq, ss, n)
                ops.gemm_mxfp8(sq, ss, mx['t_qkv'].q, mx['t_qkv'].s, mx['t_qkv'].b, side)
                ops.qkv_time_attention2_mx(xq, xs, mx['t_qkv'].q, mx['t_qkv'].s, mx['t_qkv'].b, side, aq, part, n_seq=n, scale=0.125, out_scales=as_)
                ops.attention_cls_combine_mx(part, aq, as_, n_part=33, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
            elif fuse_time:
                cq.copy_(xq.view(n, VIS_L, D)[:, 0])
                cs[:, :n].copy_(xs[:, :rows].view(6, n, VIS_L, 4)[:, :, 0])
                ops.gemm_mxfp8(cq, cs, mx['t_qkv'].q, mx['t_qkv'].s, mx['t_qkv'].b, qkv_cls)
                if fuse_attn:
                    # ... and writes the projection's MXFP8 operand itself, into a second operand buffer (other workgroups still read xq / xs)
                    ops.qkv_time_attention_mx(xq, xs, mx['t_qkv'].q, mx['t_qkv'].s, mx['t_qkv'].b, qkv_cls, aq, part, n_seq=n, n_groups=196, scale=0.125, out_scales=as_)
                    ops.attention_cls_combine_mx(part, aq, as_, n_part=49, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
                else:
                    ops.qkv_time_attention_mx(xq, xs, mx['t_qkv'].q, mx['t_qkv'].s, mx['t_qkv'].b, qkv_cls, xn, part, n_seq=n, n_groups=196, scale=0.125)
                    ops.attention_cls_combine(part, xn, n_part=49, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
            else:
                ops.gemm_mxfp8(xq, xs, mx['t_qkv'].q, mx['t_qkv'].s, mx['t_qkv'].b, qkv)
                divided('time')
            if fuse_time2 and proj_bf16:
                pass                                                       # (projected and re-quantised above)
            else:
                if fuse_time and fuse_attn:
                    tq, ts = aq, as_
                else:
                    ops.quantize_mxfp8(xn, xq, xs)
                    tq, ts = xq, xs
                if fuse:
                    ops.gemm_mx_res_ln(tq, ts, mx['t_proj'].q, mx['t_proj'].s, mx['t_proj'].b, X, b['norm1'].g, b['norm1'].b, xq, xs, EPS_VIS)
                else:
                    ops.gemm_mxfp8(tq, ts, mx['t_proj'].q, mx['t_proj'].s, mx['t_proj'].b, X, residual=X)
                    ops.layernorm_mxfp8(X, b['norm1'].g, b['norm1'].b, xq, xs, EPS_VIS)
            if fuse_space and proj_bf16:
                ops.space_side_rows_mx(xq, xs, sq, ss, n)
                ops.gemm_mxfp8(sq, ss, mx['s_qkv'].q, mx['s_qkv'].s, mx['s_qkv'].b, side)
                ops.qkv_space_attention_mx(xq, xs, mx['s_qkv'].q, mx['s_qkv'].s, mx['s_qkv'].b, side, att_b, part, n_seq=n, scale=0.125)
                ops.attention_cls_combine(part, att_b, n_part=8, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
                ops.gemm_res_ln(att_b, b['s_proj'].wk, b['s_proj'].b, X, b['norm2'].g, b['norm2'].b, xn, EPS_VIS)
                ops.quantize_mxfp8(xn, xq, xs)
            elif fuse_space:
                ops.space_side_rows_mx(xq, xs, sq, ss, n)
                ops.gemm_mxfp8(sq, ss, mx['s_qkv'].q, mx['s_qkv'].s, mx['s_qkv'].b, side)
                ops.qkv_space_attention_mx(xq, xs, mx['s_qkv'].q, mx['s_qkv'].s, mx['s_qkv'].b, side, aq, part, n_seq=n, scale=0.125, out_scales=as_)
                ops.attention_cls_combine_mx(part, aq, as_, n_part=8, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
                pq, ps = aq, as_
            else:
                ops.gemm_mxfp8(xq, xs, mx['s_qkv'].q, mx['s_qkv'].s, mx['s_qkv'].b, qkv)
                pq, ps = xq, xs
            if fuse_space:
                pass
            elif fuse_attn:
                # the space attention writes the projection's MXFP8 operand itself (sf_attention_cls_partial_mx): no bf16 output, no quantisation pass
                ops.attention_cls_partial_mx(q, k, v, xq, xs, part, n_seq=n, seq_rows=VIS_L, n_groups=8, row0=1, group_stride=196, tok_stride=1, n_tok=196,
                                             cls_row=0, heads=12, scale=0.125)
                ops.attention_cls_combine_mx(part, xq, xs, n_part=8, n_seq=n, out_seq_rows=VIS_L, out_row=0, heads=12)
            else:
                divided('space')
                ops.quantize_mxfp8(xn, xq, xs)
            if fuse_space and proj_bf16:
                pass
            elif fuse:
                ops.gemm_mx_res_ln(pq, ps, mx['s_proj'].q, mx['s_proj'].s, mx['s_proj'].b, X, b['norm2'].g, b['norm2'].b, xq, xs, EPS_VIS)
            else:
                ops.gemm_mxfp8(pq, ps, mx['s_proj'].q, mx['s_proj'].s, mx['s_proj'].b, X, residual=X)
                ops.layernorm_mxfp8(X, b['norm2'].g, b['norm2'].b, xq, xs, EPS_VIS)
            if fc2_bf16:
                ops.gemm_mxfp8(xq, xs, mx['fc1'].q, mx['fc1'].s, mx['fc1'].b, hid_b, gelu=True)            # bf16 hidden
                if bi + 1 < nb:
                    nx = self.v_blocks[bi + 1]['norm3']
                    ops.gemm_res_ln(hid_b, b['fc2'].wk, b['fc2'].b, X, nx.g, nx.b, xn, EPS_VIS)
                    ops.quantize_mxfp8(xn, xq, xs)
                else:
                    ops.gemm(hid_b, b['fc2'].w, b['fc2'].b, X, residual=X)
                continue
            ops.gemm_mxfp8(xq, xs, mx['fc1'].q, mx['fc1'].s, mx['fc1'].b, hq, gelu=True, out_scales=hs)
            if fuse and bi + 1 < nb:
                nx = self.v_blocks[bi + 1]['norm3']
                ops.gemm_mx_res_ln(hq, hs, mx['fc2'].q, mx['fc2'].s, mx['fc2'].b, X, nx.g, nx.b, xq, xs, EPS_VIS)
            else:
                ops.gemm_mxfp8(hq, hs, mx['fc2'].q, mx['fc2'].s, mx['fc2'].b, X, residual=X)

    def _visual_tail(self, X, n, out, tok_keep):
        # drop CLS -> final norm -> per-frame sequences with the aggregator CLS in front (mf:231-232, 356-375)
        Z = self._buf('Z', n * 8 * AGG_V * D, torch.float32).view(n * 8 * AGG_V, D)
        ops.broadcast_rows(Z, self.v_agg['cls'], n_seq=n * 8, dst_seq_rows=AGG_V)
        ops.layernorm(X, self.v_norm.g, self.v_norm.b, Z, EPS_VIS, rows=n * VIS_P,
                      in_map=ops.rowmap(VIS_P, VIS_P, VIS_L, 0, 1, 1), out_map=ops.rowmap(VIS_P, 196, 8 * AGG_V, AGG_V, 1, 1))
        zkeep = None
        if tok_keep is not None:                                        # per-frame key mask [agg cls = keep; 196 patches] (motionformer.py:237-243, 308-317)
            zkeep = torch.ones(n * 8, AGG_V, device=self.dev, dtype=torch.uint8)
            zkeep[:, 1:] = tok_keep.view(n, VIS_L)[:, 1:].reshape(n * 8, 196)
            zkeep = zkeep.reshape(-1)
        self._agg_layer(Z, n * 8, AGG_V, self.v_agg, out, 'vagg', key_keep=zkeep)

    def extract_vfeats(self, vis: torch.Tensor, vis_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """vis (B, S, Tv=16, C=3, H, W) -> (B, S, 8, 768) fp32 (Synchformer.extract_vfeats, sync_model.py:72-80).
        vis_mask: optional bool tensor shaped like vis, False = masked content (token masks, sync_model.py:75-76)."""
        B, S = vis.shape[:2]
        vid = vis.reshape(B * S, *vis.shape[2:])
        if not vid.is_contiguous():
            vid = vid.contiguous()
        keep = None
        if vis_mask is not None:
            if vis_mask.shape != vis.shape:
                raise ValueError(f'vis_mask {tuple(vis_mask.shape)} must have the shape of vis {tuple(vis.shape)}')
            keep = vis_mask.to(self.dev).to(torch.bool).reshape(vid.shape).contiguous()
        out = torch.empty(B * S * 8, D, device=self.dev, dtype=torch.float32)
        split_on = self.vis_split_mode == 'always' or (self.vis_split_mode == 'graph' and self._in_capture)
        if split_on and max(2, self.vis_split_min) <= B * S <= self.vis_split_max and self.capture_blocks is None:      # (the tests' per-block capture wants one chunk)
            n, k = B * S, self.vis_split_parts
            cuts = [(n * i + k - 1) // k for i in range(k + 1)]
            self._parts([(lambda lo=lo, hi=hi: self._visual_chunk(vid[lo:hi], out[lo * 8:hi * 8], keep=None if keep is None else keep[lo:hi]))
                         for lo, hi in zip(cuts[:-1], cuts[1:]) if hi > lo])
            return out.view(B, S, 8, D)
        for s0 in range(0, B * S, self.seg_chunk):
            n = min(self.seg_chunk, B * S - s0)
            self._visual_chunk(vid[s0:s0 + n], out[s0 * 8:(s0 + n) * 8], keep=None if keep is None else keep[s0:s0 + n])
        return out.view(B, S, 8, D)

    def _parts(self, fns):
        """Run independent sub-schedules side by side: fns[0] on the current stream, the others on the engine's further visual streams with workspaces of their own
        (`_ws_tag`); fork / join by events, so the set captures into a HIP graph like any other part of the forward."""
        if self._v_side is None:
            self._v_side, self._v_fork = [], torch.cuda.Event()
        while len(self._v_side) < len(fns) - 1:
            self._v_side.append((torch.cuda.Stream(device=self.dev), torch.cuda.Event()))
        main = torch.cuda.current_stream()
        self._v_fork.record(main)
        for i, fn in enumerate(fns[1:]):
            side, join = self._v_side[i]
            with torch.cuda.stream(side):
                side.wait_event(self._v_fork)
                self._ws_tag = f'h{i + 1}:'
                try:
                    fn()
                finally:
                    self._ws_tag = ''
                join.record(side)
        fns[0]()
        for i in range(len(fns) - 1):
            main.wait_event(self._v_side[i][1])

    # ------------------------------------------------------------------------------------------------
    # audio branch
    # ------------------------------------------------------------------------------------------------
    def extract_afeats(self, aud: torch.Tensor, aud_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """aud (B, S, 1, F=128, Ta=66) fp32 -> (B, S, 6, 768) fp32 (sync_model.py:82-89, ast.py:137-201).
        aud_mask: optional bool tensor shaped like aud, False = masked content (sync_model.py:85-86)."""
        B, S, _, Fa, Ta = aud.shape
        n = B * S
        spec = aud.reshape(n, Fa, Ta).to(torch.float32)
        if not spec.is_contiguous():
            spec = spec.contiguous()
        nf, nt = (Fa - 16) // 10 + 1, (Ta - 16) // 10 + 1
        P, L = nf * nt, nf * nt + 2
        if L != self.a_table.shape[0]:
            raise ValueError(f'spectrogram gives {L} tokens but position table has {self.a_table.shape[0]}')
        rows = n * L
        X = self._buf('Xa', rows * D, torch.float32).view(rows, D)
        agg_rows = n * nt * (nf + 1)
        xn = self._buf('XNa', max(rows, agg_rows) * D, torch.bfloat16)[:rows * D].view(rows, D)      # the audio branch has its own workspaces:
        big = self._buf('BIGa', max(rows, agg_rows) * FF, torch.bfloat16)                              # it runs next to the visual one (both_towers)
        patches = big[:n * P * 256].view(n * P, 256)
        ops.im2col_spec(spec, patches)
        ops.broadcast_rows(X, self.a_table, n_seq=n, dst_seq_rows=L)
        tokmap = ops.rowmap(P, P, L, 0, 1, 2)
        ops.gemm(patches, self.a_pe.w, self.a_pe.b, X, residual=X, c_map=tokmap, r_map=tokmap)

        tok_keep = None
        if aud_mask is not None:
            keep = aud_mask.to(self.dev).to(torch.bool).reshape(n, Fa, Ta)
            tok_keep = ops.token_mask_spec(keep, self.a_w0_sign, torch.empty(rows, device=self.dev, dtype=torch.uint8))

        def full_attn(q3, o):
            ops.attention(q3[:, :D], q3[:, D:2 * D], q3[:, 2 * D:], o, n_seq=n, seq_rows=L, n_groups=1, row0=0,
                          group_stride=0, tok_stride=1, n_tok=L, cls_row=-1, heads=12, head_dim=64, scale=0.125, key_keep=tok_keep)
        for ly in self.a_layers:   # ASTLayer.forward (modeling_ast.py:294-322)
            self._encoder_layer(X, rows, xn, big, ly['ln1'], ly['qkv'], full_attn, ly['o'], ly['ln2'], ly['fc1'], ly['fc2'],
                                EPS_AST)
        # final layernorm, drop CLS/DISTILL, regroup (fi, ti) -> per-time-step sequences (ast.py:232-236, 265-266)
        La = nf + 1
        Z = self._buf('Za', agg_rows * D, torch.float32).view(agg_rows, D)
        ops.broadcast_rows(Z, self.a_agg['cls'], n_seq=n * nt, dst_seq_rows=La)
        ops.layernorm(X, self.a_norm.g, self.a_norm.b, Z, EPS_AST, rows=n * P, in_map=ops.rowmap(P, P, L, 0, 1, 2),
                      out_map=ops.rowmap(P, nt, nt * La, 1, La, 1))
        out = torch.empty(n * nt, D, device=self.dev, dtype=torch.float32)
        zkeep = None
        if tok_keep is not None:                                        # per-time-step key mask [agg cls = keep; 12 frequency tokens] (ast.py:188-193, 269-271)
            zkeep = torch.ones(n * nt, La, device=self.dev, dtype=torch.uint8)
            zkeep[:, 1:] = tok_keep.view(n, L)[:, 2:].reshape(n, nf, nt).transpose(1, 2).reshape(n * nt, nf)
            zkeep = zkeep.reshape(-1)
        self._agg_layer(Z, n * nt, La, self.a_agg, out, 'aagg', key_keep=zkeep)
        return out.view(B, S, nt, D)

    # ------------------------------------------------------------------------------------------------
    # Stage-1 segment-level contrastive head (AVCLIP, train_clip_src/open_clip/model.py:449-585)
    # ------------------------------------------------------------------------------------------------
    def pool_segments(self, feat: torch.Tensor, normalize: bool = False) -> torch.Tensor:
        """(B, S, t, 768) aggregator outputs -> (B*S, 768): AveragePooling 'BS t D -> BS D' (motionformer.py:139, ast.py:88),
        optionally followed by F.normalize (AVCLIP.encode_stream, open_clip/model.py:522-533)."""
        B, S, t = feat.shape[:3]
        f2 = feat.reshape(B * S * t, D)
        if not f2.is_contiguous():
            f2 = f2.contiguous()
        out = torch.empty(B * S, D, device=self.dev, dtype=torch.float32)
        return ops.meanpool_l2norm(f2, out, t, normalize)

    def l2_normalize(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(x)
        return ops.meanpool_l2norm(x.contiguous(), out, 1, True)

    def contrastive_loss(self, vfeat: torch.Tensor, afeat: torch.Tensor, vfeat_all: torch.Tensor, afeat_all: torch.Tensor,
                         logit_scale: float, row_offset: int = 0):
        """AVCLIP.compute_loss (open_clip/model.py:506-525) with alpha = 0: sim_v2a = vfeat @ afeat_all^T / scale (and a2v),
        targets = eye(n, m) i.e. class `row_offset + i` for row i, loss = (CE(v2a) + CE(a2v)) / 2.
        NOTE the reference's eye(n, m) puts the positive of LOCAL row i at column i even when features were gathered
        from other ranks (row_offset stays 0 there, open_clip/model.py:516); `row_offset` is only for callers who want
        the rank-aware diagonal.  Returns (loss (1,) fp32, sim_v2a, sim_a2v)."""
        n, m = vfeat.shape[0], afeat_all.shape[0]
        sim_v2a = torch.empty(n, m, device=self.dev, dtype=torch.float32)
        sim_a2v = torch.empty(n, vfeat_all.shape[0], device=self.dev, dtype=torch.float32)
        ops.similarity(vfeat, afeat_all, sim_v2a, 1.0 / logit_scale)
        ops.similarity(afeat, vfeat_all, sim_a2v, 1.0 / logit_scale)
        tgt = torch.arange(row_offset, row_offset + n, device=self.dev, dtype=torch.int64)
        losses = torch.empty(2, device=self.dev, dtype=torch.float32)
        ops.cross_entropy(sim_v2a, tgt, losses[0:1])
        ops.cross_entropy(sim_a2v, tgt, losses[1:2])
        return losses, sim_v2a, sim_a2v

    # ------------------------------------------------------------------------------------------------
    # sync transformer + top level
    # ------------------------------------------------------------------------------------------------
    def project(self, feat: torch.Tensor, which: str) -> torch.Tensor:
        """vproj / aproj (sync_model.py:55-56): (B, S, t, 768) fp32 -> (B, S*t, 768) fp32."""
        proj = self.vproj if which == 'v' else self.aproj
        B, n_tok = feat.shape[0], feat.shape[1] * feat.shape[2]
        f2 = feat.reshape(B * n_tok, D)
        fb = self._buf('proj_in', B * n_tok * D, torch.bfloat16).view(B * n_tok, D)
        ops.gather_rows(f2, fb, B * n_tok)
        out = torch.empty(B * n_tok, D, device=self.dev, dtype=torch.float32)
        ops.gemm(fb, proj.w, proj.b, out)
        return out.view(B, n_tok, D)

    def global_transformer(self, v: torch.Tensor, a: torch.Tensor, apply_head: bool = True) -> torch.Tensor:
        """GlobalTransformer.forward (sync_model.py:150-173).  v (B, Sv, 768), a (B, Sa, 768) fp32 on device (already
        projected) -> logits fp32 (B, n_out); apply_head=False (`attempt_to_apply_heads=False`, :170-172): ln_f of every token, fp32 (B, L, 768)."""
        B, Sv, Sa = v.shape[0], v.shape[1], a.shape[1]
        L = 2 + Sv + Sa
        table = self._sync_table(Sv, Sa)
        rows = B * L
        X = self._buf('Xs', rows * D, torch.float32).view(rows, D)
        xn = self._buf('XNs', rows * D, torch.bfloat16).view(rows, D)
        big = self._buf('BIGs', rows * FF, torch.bfloat16)
        ops.broadcast_rows(X, table, n_seq=B, dst_seq_rows=L)
        for feat, ln, n_tok, off in ((v, self.s_vln, Sv, 1), (a, self.s_aln, Sa, 2 + Sv)):
            f2 = feat.reshape(B * n_tok, D)
            if not f2.is_contiguous():
                f2 = f2.contiguous()
            ops.layernorm(f2, ln.g, ln.b, X, EPS_SYNC, out_map=ops.rowmap(n_tok, n_tok, L, 0, 1, off), accumulate=True)
        hd = D // self.s_heads

        def full_attn(q3, o):
            ops.attention(q3[:, :D], q3[:, D:2 * D], q3[:, 2 * D:], o, n_seq=B, seq_rows=L, n_groups=1, row0=0,
                          group_stride=0, tok_stride=1, n_tok=L, cls_row=-1, heads=self.s_heads, head_dim=hd,
                          scale=1.0 / math.sqrt(hd))
        for b in self.s_blocks:   # Block.forward (modules/transformer.py:93-97)
            self._encoder_layer(X, rows, xn, big, b['ln1'], b['qkv'], full_attn, b['proj'], b['ln2'], b['fc1'], b['fc2'],
                                EPS_SYNC)
        if not apply_head:
            tokens = torch.empty(rows, D, device=self.dev, dtype=torch.float32)
            ops.layernorm(X, self.s_lnf.g, self.s_lnf.b, tokens, EPS_SYNC, rows=rows)
            return tokens.view(B, L, D)
        cls = xn[:B]
        ops.layernorm(X, self.s_lnf.g, self.s_lnf.b, cls, EPS_SYNC, rows=B, in_map=ops.rowmap(1, 1, L, 0, 0, 0))
        logits = torch.empty(B, self.n_out, device=self.dev, dtype=torch.float32)
        ops.gemm(cls, self.s_head.w, self.s_head.b, logits, M=B)
        return logits

    def sync_transformer(self, vfeat: torch.Tensor, afeat: torch.Tensor) -> torch.Tensor:
        """vproj/aproj + GlobalTransformer: segment features (B,S,tv,768) / (B,S,ta,768) fp32 -> logits (B, n_out)."""
        return self.global_transformer(self.project(vfeat, 'v'), self.project(afeat, 'a'))

    def forward(self, vis: torch.Tensor, aud: torch.Tensor, vis_mask: Optional[torch.Tensor] = None,
                aud_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Synchformer.forward (sync_model.py:38-70) without the loss: logits (B, n_out) fp32."""
        return self.sync_transformer(*self.both_towers(lambda: self.extract_vfeats(vis, vis_mask), aud, aud_mask))

    def both_towers(self, visual_fn, aud: torch.Tensor, aud_mask: Optional[torch.Tensor] = None):
        """(vfeats, afeats) with the audio tower on a second HIP stream NEXT TO the visual tower.  The two are independent until the sync
        transformer (sync_model.py:45-52); the audio tower is 3.4 % of the FLOPs in small, latency-bound launches (16.5k token rows) that fit
        into the gaps and the HBM-bound phases of the visual tower: +1.2 % clips/s (128.7 -> 130.3, interleaved A/B on one box).
        SF_AUDIO_SIDE_STREAM=0 runs them back to back on the current stream."""
        if not self.audio_side_stream:
            return visual_fn(), self.extract_afeats(aud, aud_mask)
        if self._a_side is None:
            self._a_side, self._a_fork, self._a_join = torch.cuda.Stream(device=self.dev), torch.cuda.Event(), torch.cuda.Event()
        main = torch.cuda.current_stream()
        self._a_fork.record(main)
        with torch.cuda.stream(self._a_side):
            self._a_side.wait_event(self._a_fork)                      # inputs were produced on the caller's stream
            af = self.extract_afeats(aud, aud_mask)
            self._a_join.record(self._a_side)
        vf = visual_fn()
        main.wait_event(self._a_join)
        af.record_stream(main)                                          # allocated under the side stream, consumed on the caller's
        return vf, af

    # ------------------------------------------------------------------------------------------------
    # whole clips in, segmenting on the device
    # ------------------------------------------------------------------------------------------------
    def extract_vfeats_clips(self, frames: torch.Tensor, v_start: int, v_stride: int, n_seg: int) -> torch.Tensor:
        """frames (B, T, 3, 224, 224) u8|float on device -> (B, n_seg, 8, 768): segment s = frames [v_start + s*v_stride, +16), read
        straight from the clip by the patch gather (the 50 %-overlapping segments are never materialised)."""
        B = frames.shape[0]
        frames = frames.contiguous()
        out = torch.empty(B * n_seg * 8, D, device=self.dev, dtype=torch.float32)
        per = max(1, self.seg_chunk // n_seg)                                 # clips per chunk
        for b0 in range(0, B, per):
            nb = min(per, B - b0)
            self._visual_chunk(frames[b0:b0 + nb], out[b0 * n_seg * 8:(b0 + nb) * n_seg * 8], clip_seg=(v_start, v_stride, n_seg))
        return out.view(B, n_seg, 8, D)

    def forward_clips(self, frames: torch.Tensor, wave: torch.Tensor, mel, v_fps: int = 25, a_fps: int = 16000, n_segments: int = 14,
                      segment_size_vframes: int = 16, step_size_seg: float = 0.5) -> torch.Tensor:
        """Un-segmented clips -> logits: frames (B, T, 3, 224, 224) uint8 (after the spatial crop), wave (B, n_samples) fp32 16 kHz,
        `mel` a synchformer_amd.frontend.MelFrontend.  Equivalent to GenerateMultipleSegments -> RGBToHalfToZeroOne -> RGBNormalize ->
        AudioMelSpectrogram -> AudioLog -> PadOrTruncate -> AudioNormalizeAST -> PermuteStreams -> Synchformer.forward
        (configs/sync.yaml:222-249, sync_model.py:38-70) with is_start_random False."""
        from .frontend import segment_ranges
        r = segment_ranges(frames.shape[1], wave.shape[1], v_fps, a_fps, segment_size_vframes, n_segments, step_size_seg)
        aud = mel.segments(wave, r['a_start'], r['a_stride'], r['n_segments'], r['a_size'])
        return self.sync_transformer(*self.both_towers(lambda: self.extract_vfeats_clips(frames, r['v_start'], r['v_stride'], r['n_segments']), aud))

    # ------------------------------------------------------------------------------------------------
    # HIP-graph replay of the whole forward (launch-bound regimes: single-clip latency, small batches)
    # ------------------------------------------------------------------------------------------------
    def capture(self, vis: torch.Tensor, aud: torch.Tensor):
        """Capture forward() for THESE input shapes / dtypes into one HIP graph (the schedule is a fixed sequence of ~900 launches
        with no host decisions in it) and return `run(vis, aud) -> logits`.  Replay removes the per-launch host cost (ctypes +
        hipLaunchKernel, ~5 us each), which is what bounds a single-clip forward.  The returned logits tensor is reused by every
        replay: copy it out before the next call if it must survive."""
        static_vis, static_aud = vis.clone(), aud.clone()
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        self._in_capture = True                                      # (the two-halves split of small batches is a graph-mode schedule: warm-up and capture both take it)
        try:
            with torch.cuda.stream(side):                            # warm-up off the capture: workspaces, lazy function attributes
                for _ in range(2):
                    self.forward(static_vis, static_aud)
            torch.cuda.current_stream(self.dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.forward(static_vis, static_aud)
        finally:
            self._in_capture = False
        generation = getattr(self, '_sync_generation', 0)

        def run(v: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
            if getattr(self, '_sync_generation', 0) != generation:
                raise RuntimeError('the sync-module weights were refreshed (load_sync_weights) after this graph was captured: its launches point at the '
                                   'released operand buffers - capture again')
            if v.shape != static_vis.shape or a.shape != static_aud.shape or v.dtype != static_vis.dtype:
                raise ValueError('captured graph serves one input shape/dtype; capture again for another')
            static_vis.copy_(v, non_blocking=True)
            static_aud.copy_(a, non_blocking=True)
            graph.replay()
            return static_out
        run.graph = graph
        return run
