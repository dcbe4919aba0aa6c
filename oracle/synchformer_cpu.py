"""CPU oracle for the Synchformer hot path - TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain fp32 restatement (torch CPU ops, functional style, own layout) of the reference algorithm behind
`Synchformer.forward()`.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this; the shipped package (`synchformer_amd/`) never does and fails loudly without its HIP library.

Pinning: `tests/test_oracle_cpu.py::test_oracle_matches_real_reference` runs this file against the REAL reference imported from
/root/reference (build container only) and `tests/golden/*.npz` hold outputs of the real reference on
seeded inputs/weights (`tests/golden/make_golden.py`), so the oracle is pinned both ways.  The mel
front-end (`mel_frontend`) restates torchaudio's MelSpectrogram, which is absent from /root/reference and
this image -> that one function is "parity unpinned" (SURVEY.md §8c).

Every function cites the reference file:line it follows (paths relative to /root/reference).
All tensors fp32; `sd` is a flat dict with the reference's state-dict keys.
"""
import math

import torch
import torch.nn.functional as F

EPS_VIS = 1e-6    # motionformer_src/video_model_builder.py:39, motionformer.py:126, ast.py:74
EPS_AST = 1e-12   # transformers ASTConfig.layer_norm_eps (modeling_ast.py:291-292)
EPS_SYNC = 1e-5   # torch.nn.LayerNorm default (sync_model.py:125-126,142; modules/transformer.py:84-85)


def _ln(x, sd, name, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + '.weight'], sd[name + '.bias'], eps)


def _lin(x, sd, name):
    return F.linear(x, sd[name + '.weight'], sd[name + '.bias'])


def _gelu(x):
    # exact erf GELU everywhere: nn.GELU() default / ACT2FN['gelu'] (vit_helper.py:382, modeling_ast.py:252-255)
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def _softmax_attn(q, k, v, scale, key_keep=None):
    """softmax(q k^T * scale) v over the last two dims; q (..., Nq, d), k/v (..., Nk, d).  key_keep (..., Nk) bool broadcastable
    to the batch dims: keys with False get -inf before the softmax (qkv_attn, vit_helper.py:34-42; modeling_ast.py:160-163)."""
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if key_keep is not None:
        s = s.masked_fill(~key_keep.unsqueeze(-2), float('-inf'))
    return torch.matmul(torch.softmax(s, dim=-1), v)


def token_mask_from_content(patches_keep, w0):
    """The reference's NaN trick (video_model_builder.py:185-201, modeling_ast.py:515-530): an indicator (1 = kept, inf = masked
    content) is pushed through the patch embedding; a token is MASKED iff output channel 0 is NaN, i.e. iff the masked elements
    of its patch meet filter-0 weights of both signs (inf - inf) or a zero weight (inf * 0).  A patch whose masked elements all
    meet same-sign weights gives +-inf, which is not NaN: that token stays.
    patches_keep (N, P, K) bool content mask gathered like the patches; w0 (K,) = filter 0.  -> (N, P) bool, True = keep."""
    m = ~patches_keep
    pos = (m & (w0 > 0)).any(-1)
    neg = (m & (w0 < 0)).any(-1)
    zero = (m & (w0 == 0)).any(-1)
    return ~((pos & neg) | zero)


# ----------------------------------------------------------------------------------------------------
# Visual branch: Motionformer, divided space-time attention
# ----------------------------------------------------------------------------------------------------
def patch_embed_3d(x, w, b):
    """PatchEmbed3D (vit_helper.py:422-445): Conv3d(3->768, k=s=(2,16,16)) then flatten(2).T.
    x (N, 3, T, H, W) -> (N, T/2*H/16*W/16, 768), token order (t, h, w).  Restated as patch-gather + GEMM
    with K ordered (c, dt, dh, dw) = the Conv3d weight's own flattening."""
    N, C, T, H, W = x.shape
    pt, ph, pw = w.shape[2:]
    t, h, ww = T // pt, H // ph, W // pw
    p = x.reshape(N, C, t, pt, h, ph, ww, pw).permute(0, 2, 4, 6, 1, 3, 5, 7)
    p = p.reshape(N, t * h * ww, C * pt * ph * pw)
    return p @ w.reshape(w.shape[0], -1).t() + b


def vis_pos_table(sd, p, frames=8):
    """'separate' positional table (video_model_builder.py:248-254): row 0 = pos_embed[0]; row 1+f*196+n =
    pos_embed[1+n] + temp_embed[f]."""
    pos, temp = sd[p + '.pos_embed'][0], sd[p + '.temp_embed'][0]
    n = pos.shape[0] - 1
    body = pos[1:].unsqueeze(0) + temp[:frames].unsqueeze(1)       # (f, n, D)
    return torch.cat([pos[:1], body.reshape(frames * n, -1)], 0)   # (1 + f*n, D)


def divided_attention(x, sd, p, mode, heads=12, frames=8, tok_keep=None):
    """DividedAttention.forward + qkv_attn (vit_helper.py:100-158, :34-42).
    x (N, 1+f*n, D).  CLS query attends all tokens; patch (f, n) attends [CLS] + its group:
    mode 'time' -> same n over all f; mode 'space' -> same f over all n.  q is scaled by d^-0.5 first (:113)."""
    N, L, Dm = x.shape
    d = Dm // heads
    n = (L - 1) // frames
    qkv = _lin(x, sd, p + '.qkv').reshape(N, L, 3, heads, d).permute(2, 0, 3, 1, 4)   # (3, N, h, L, d)
    q, k, v = qkv[0] * (d ** -0.5), qkv[1], qkv[2]
    kk_all = None if tok_keep is None else tok_keep.unsqueeze(1)                       # (N, 1, L): same mask for every head (:107-110)
    out_cls = _softmax_attn(q[:, :, :1], k, v, 1.0, kk_all)                           # (N, h, 1, d)

    def grp(t):  # patches (N, h, f*n, d) -> groups
        t = t[:, :, 1:].reshape(N, heads, frames, n, d)
        return t.transpose(2, 3) if mode == 'time' else t     # time: (N,h,n,f,d)  space: (N,h,f,n,d)
    qg, kg, vg = grp(q), grp(k), grp(v)
    G = qg.shape[2]
    kc = k[:, :, :1].unsqueeze(2).expand(N, heads, G, 1, d)
    vc = v[:, :, :1].unsqueeze(2).expand(N, heads, G, 1, d)
    gk = None
    if tok_keep is not None:                                                          # mask rearranged like the keys (:136-141)
        mg = tok_keep[:, 1:].reshape(N, frames, n)
        mg = mg.transpose(1, 2) if mode == 'time' else mg                              # (N, G, T)
        gk = torch.cat([tok_keep[:, :1].unsqueeze(1).expand(N, G, 1), mg], 2).unsqueeze(1)   # (N, 1, G, 1+T)
    og = _softmax_attn(qg, torch.cat([kc, kg], 3), torch.cat([vc, vg], 3), 1.0, gk)
    if mode == 'time':
        og = og.transpose(2, 3)
    out = torch.cat([out_cls, og.reshape(N, heads, frames * n, d)], 2)                # (N, h, L, d)
    out = out.transpose(1, 2).reshape(N, L, Dm)
    return _lin(out, sd, p + '.proj')


def divided_block(x, sd, p, tok_keep=None, drop_path=None):
    """DividedSpaceTimeBlock.forward (vit_helper.py:364-376).  DropPath is identity in eval; in train mode (`drop_path` = (space, mlp) pair of
    per-sample scale vectors (N,), each 0 or 1 / keep_prob, or None) the space-attention and MLP branches are scaled per sample
    (vit_helper.py:372,375; timm DropPath: x * bernoulli(keep) / keep over dim 0) - the time-attention branch never is (:367-369)."""
    dps, dpm = drop_path if drop_path is not None else (None, None)
    x = x + divided_attention(_ln(x, sd, p + '.norm3', EPS_VIS), sd, p + '.timeattn', 'time', tok_keep=tok_keep)
    br = divided_attention(_ln(x, sd, p + '.norm1', EPS_VIS), sd, p + '.attn', 'space', tok_keep=tok_keep)
    x = x + (br if dps is None else br * dps.view(-1, 1, 1))
    h = _gelu(_lin(_ln(x, sd, p + '.norm2', EPS_VIS), sd, p + '.mlp.fc1'))           # Mlp (vit_helper.py:379-398)
    br = _lin(h, sd, p + '.mlp.fc2')
    return x + (br if dpm is None else br * dpm.view(-1, 1, 1))


def agg_encoder_layer_cls(tokens, sd, p, heads=12, keep=None):
    """BaseEncoderLayer.forward (motionformer.py:301-334) over nn.TransformerEncoderLayer(norm_first=True,
    GELU, eps 1e-6, dropout 0): prepend the aggregator's cls_token, one pre-norm encoder layer, return row 0.
    tokens (N, L, D) -> (N, D)."""
    N, L, Dm = tokens.shape
    d = Dm // heads
    z = torch.cat([sd[p + '.cls_token'].expand(N, 1, Dm), tokens], 1)
    y = _ln(z, sd, p + '.norm1', EPS_VIS)
    qkv = F.linear(y, sd[p + '.self_attn.in_proj_weight'], sd[p + '.self_attn.in_proj_bias'])
    qkv = qkv.reshape(N, L + 1, 3, heads, d).permute(2, 0, 3, 1, 4)
    kk = None                                                       # keep (N, L) bool: [cls = keep; tokens] as key mask (mf:308-317)
    if keep is not None:
        kk = torch.cat([torch.ones(N, 1, dtype=torch.bool), keep], 1).unsqueeze(1)
    a = _softmax_attn(qkv[0], qkv[1], qkv[2], d ** -0.5, kk).transpose(1, 2).reshape(N, L + 1, Dm)
    z = z + _lin(a, sd, p + '.self_attn.out_proj')
    z = z + _lin(_gelu(_lin(_ln(z, sd, p + '.norm2', EPS_VIS), sd, p + '.linear1')), sd, p + '.linear2')
    return z[:, 0]


def motionformer_segments(x, sd, p='vfeat_extractor', depth=None, cont_keep=None, drop_path=None):
    """MotionFormer.forward_segments (motionformer.py:225-252) with forward_features
    (video_model_builder.py:174-274).  x (N, 3, 16, 224, 224) -> (N, 8, 768)."""
    N = x.shape[0]
    tok = patch_embed_3d(x, sd[p + '.patch_embed_3d.proj.weight'], sd[p + '.patch_embed_3d.proj.bias'])
    tok_keep = None
    if cont_keep is not None:                                          # cont_keep (N, 3, 16, 224, 224) bool, True = kept content
        w = sd[p + '.patch_embed_3d.proj.weight']
        pt, ph, pw = w.shape[2:]
        C, T, H, W = cont_keep.shape[1:]
        pk = cont_keep.reshape(N, C, T // pt, pt, H // ph, ph, W // pw, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(N, -1, C * pt * ph * pw)
        tok_keep = torch.cat([torch.ones(N, 1, dtype=torch.bool), token_mask_from_content(pk, w[0].reshape(-1))], 1)   # CLS kept (vmb:223-225)
    x = torch.cat([sd[p + '.cls_token'].expand(N, 1, -1), tok], 1) + vis_pos_table(sd, p)
    i = 0
    while f'{p}.blocks.{i}.norm1.weight' in sd and (depth is None or i < depth):
        x = divided_block(x, sd, f'{p}.blocks.{i}', tok_keep, None if drop_path is None else drop_path[i])
        i += 1
    x = _ln(x[:, 1:], sd, p + '.norm', EPS_VIS)                       # drop CLS, final norm (mf:231-232)
    frames = 8
    per_frame = x.reshape(N * frames, x.shape[1] // frames, -1)       # '(BS t) (h w) D' (mf:361)
    keep = None if tok_keep is None else tok_keep[:, 1:].reshape(N * frames, -1)      # (mf:237-243, 365-366)
    return agg_encoder_layer_cls(per_frame, sd, p + '.spatial_attn_agg', keep=keep).reshape(N, frames, -1)


# ----------------------------------------------------------------------------------------------------
# Audio branch: AST
# ----------------------------------------------------------------------------------------------------
def ast_patch_embed(x, w, b, stride=10):
    """ASTPatchEmbeddings (modeling_ast.py:96-117): x (N, Ta, F) -> (N,1,F,Ta) -> Conv2d(1->768, k16, s10) ->
    (N, 12*6, 768), token order (f, t).  Restated as overlapping-window gather + GEMM, K ordered (df, dt)."""
    xs = x.transpose(1, 2)                                            # (N, F, Ta)
    win = xs.unfold(1, 16, stride).unfold(2, 16, stride)             # (N, nf, nt, 16, 16)
    N, nf, nt = win.shape[:3]
    return win.reshape(N, nf * nt, 256) @ w.reshape(w.shape[0], 256).t() + b


def ast_layer(x, sd, p, heads=12, tok_keep=None):
    """ASTLayer.forward (modeling_ast.py:294-322) with ASTSelfAttention (:139-184): scores / sqrt(d)."""
    N, L, Dm = x.shape
    d = Dm // heads
    y = _ln(x, sd, p + '.layernorm_before', EPS_AST)

    def hd(name):
        return _lin(y, sd, f'{p}.attention.attention.{name}').reshape(N, L, heads, d).transpose(1, 2)
    a = _softmax_attn(hd('query'), hd('key'), hd('value'), 1.0 / math.sqrt(d), None if tok_keep is None else tok_keep.unsqueeze(1))
    h = x + _lin(a.transpose(1, 2).reshape(N, L, Dm), sd, p + '.attention.output.dense')
    m = _gelu(_lin(_ln(h, sd, p + '.layernorm_after', EPS_AST), sd, p + '.intermediate.dense'))
    return h + _lin(m, sd, p + '.output.dense')


def ast_segments(x, sd, p='afeat_extractor', depth=None, cont_keep=None):
    """AST.forward_segments (ast.py:178-201) with ASTModel.forward (modeling_ast.py:488-555).
    x (N, Ta=66, F=128) -> (N, 6, 768)."""
    N = x.shape[0]
    e = p + '.ast.embeddings'
    tok = ast_patch_embed(x, sd[e + '.patch_embeddings.projection.weight'], sd[e + '.patch_embeddings.projection.bias'])
    x = torch.cat([sd[e + '.cls_token'].expand(N, 1, -1), sd[e + '.distillation_token'].expand(N, 1, -1), tok], 1)
    x = x + sd[e + '.position_embeddings'][:, :x.shape[1]]
    tok_keep = None
    if cont_keep is not None:                                          # cont_keep (N, Ta, F) bool (ast.py:142, modeling_ast.py:515-530)
        w = sd[e + '.patch_embeddings.projection.weight']
        win = cont_keep.transpose(1, 2).unfold(1, 16, 10).unfold(2, 16, 10)          # (N, nf, nt, 16, 16) like ast_patch_embed
        pk = win.reshape(N, -1, 256)
        tok_keep = torch.cat([torch.ones(N, 2, dtype=torch.bool), token_mask_from_content(pk, w[0].reshape(-1))], 1)   # CLS, DISTILL kept
    i = 0
    while f'{p}.ast.encoder.layer.{i}.layernorm_before.weight' in sd and (depth is None or i < depth):
        x = ast_layer(x, sd, f'{p}.ast.encoder.layer.{i}', tok_keep=tok_keep)
        i += 1
    x = _ln(x, sd, p + '.ast.layernorm', EPS_AST)[:, 2:]             # drop CLS+DISTILL (ast.py:232-233)
    nf, nt = 12, x.shape[1] // 12
    per_t = x.reshape(N, nf, nt, -1).transpose(1, 2).reshape(N * nt, nf, -1)   # (BS*t, f, D) (ast.py:265-266)
    keep = None if tok_keep is None else tok_keep[:, 2:].reshape(N, nf, nt).transpose(1, 2).reshape(N * nt, nf)   # (ast.py:188-193, 269-271)
    return agg_encoder_layer_cls(per_t, sd, p + '.freq_attn_agg', keep=keep).reshape(N, nt, -1)


# ----------------------------------------------------------------------------------------------------
# Sync transformer + top level
# ----------------------------------------------------------------------------------------------------
def sync_block(x, sd, p, heads=8, masks=None):
    """Block / SelfAttention (modules/transformer.py:79-97, :31-76).  `masks` (tests only) = explicit dropout multipliers
    {'attn': (N,h,L,L), 'proj': (N,L,D), 'mlp': (N,L,D)} (already scaled by 1/(1-p)) standing in for attn_drop / resid_drop /
    the MLP's Dropout of train mode (:70, :73, :90); None = eval mode."""
    N, L, Dm = x.shape
    d = Dm // heads
    masks = masks or {}
    y = _ln(x, sd, p + '.ln1', EPS_SYNC)

    def hd(name):
        return _lin(y, sd, f'{p}.attn.{name}').reshape(N, L, heads, d).transpose(1, 2)
    att = torch.softmax(torch.matmul(hd('query'), hd('key').transpose(-1, -2)) * (1.0 / math.sqrt(d)), dim=-1)
    if 'attn' in masks:
        att = att * masks['attn']
    a = torch.matmul(att, hd('value'))
    br = _lin(a.transpose(1, 2).reshape(N, L, Dm), sd, p + '.attn.proj')
    x = x + (br * masks['proj'] if 'proj' in masks else br)
    br = _lin(_gelu(_lin(_ln(x, sd, p + '.ln2', EPS_SYNC), sd, p + '.mlp.0')), sd, p + '.mlp.2')
    return x + (br * masks['mlp'] if 'mlp' in masks else br)


def global_transformer(v, a, sd, p='transformer', apply_head=True, masks=None):
    """GlobalTransformer.forward (sync_model.py:150-173).  v (B, Sv, D), a (B, Sa, D) -> logits (B, n_cls)
    (or the ln_f output (B, 1+Sv+1+Sa, D) when apply_head is False)."""
    B = v.shape[0]
    v, a = _ln(v, sd, p + '.vis_in_lnorm', EPS_SYNC), _ln(a, sd, p + '.aud_in_lnorm', EPS_SYNC)
    masks = masks or {}
    if 'tok_v' in masks:                                              # tok_drop_vis / tok_drop_aud in train mode: Dropout1d on (B, S, D) = whole tokens (sync_model.py:131-134, 160-161);
        v, a = v * masks['tok_v'], a * masks['tok_a']                 # explicit multipliers (B, S, 1), already scaled by 1 / (1 - p)
    x = torch.cat([sd[p + '.OFF_tok'].expand(B, 1, -1), v, sd[p + '.MOD_tok'].expand(B, 1, -1), a], 1)
    x = x + sd[p + '.pos_emb_cfg.pos_emb'][:, :x.shape[1]]
    if 'embd' in masks:                                               # self.drop(x) in train mode (sync_model.py:166)
        x = x * masks['embd']
    i = 0
    while f'{p}.blocks.{i}.ln1.weight' in sd:
        x = sync_block(x, sd, f'{p}.blocks.{i}', masks=masks.get(i))
        i += 1
    x = _ln(x, sd, p + '.ln_f', EPS_SYNC)
    if not apply_head:
        return x
    head = 'off_head' if p + '.off_head.weight' in sd else 'sync_head'   # sync_model.py:176-190
    return _lin(x[:, 0], sd, f'{p}.{head}')


def extract_vfeats(vis, sd, chunk=None, vis_mask=None, drop_path=None):
    """Synchformer.extract_vfeats (sync_model.py:72-80) + MotionFormer.forward (motionformer.py:182-223).
    vis (B, S, Tv, C, H, W) -> (B, S, 8, 768).  `chunk` bounds CPU memory (segments per pass); results are
    identical to one pass (the reference's own for_loop switch, motionformer.py:200-207)."""
    B, S = vis.shape[:2]
    x = vis.permute(0, 1, 3, 2, 4, 5).reshape(B * S, vis.shape[3], vis.shape[2], *vis.shape[4:]).float()
    chunk = chunk or x.shape[0]
    m = None                                                           # vis_mask: same shape as vis, True = kept (sync_model.py:75-76)
    if vis_mask is not None:
        m = vis_mask.permute(0, 1, 3, 2, 4, 5).reshape(x.shape).bool()
    # drop_path: per visual block a (space, mlp) pair of per-segment scale vectors (B*S,) or None - the Stage-1 TRAIN-mode forward
    dp = lambda i: None if drop_path is None else [tuple(None if t is None else t[i:i + chunk] for t in pair) for pair in drop_path]
    out = torch.cat([motionformer_segments(x[i:i + chunk], sd, cont_keep=None if m is None else m[i:i + chunk], drop_path=dp(i))
                     for i in range(0, x.shape[0], chunk)], 0)
    return out.reshape(B, S, *out.shape[1:])


def extract_afeats(aud, sd, aud_mask=None):
    """Synchformer.extract_afeats (sync_model.py:82-89) + AST.forward (ast.py:137-176).
    aud (B, S, 1, F, Ta) -> (B, S, 6, 768)."""
    B, S, _, Fa, Ta = aud.shape
    x = aud.reshape(B * S, Fa, Ta).transpose(1, 2).float()            # (BS, Ta, F)
    m = None if aud_mask is None else aud_mask.reshape(B * S, Fa, Ta).transpose(1, 2).bool()     # (sync_model.py:85-86)
    out = ast_segments(x, sd, cont_keep=m)
    return out.reshape(B, S, *out.shape[1:])


def synchformer_forward(sd, vis, aud, targets=None, chunk=None, vis_mask=None, aud_mask=None):
    """Synchformer.forward (sync_model.py:38-70) -> (loss | None, logits)."""
    v = _lin(extract_vfeats(vis, sd, chunk, vis_mask), sd, 'vproj')
    a = _lin(extract_afeats(aud, sd, aud_mask), sd, 'aproj')
    B = v.shape[0]
    logits = global_transformer(v.reshape(B, -1, v.shape[-1]), a.reshape(B, -1, a.shape[-1]), sd)
    loss = F.cross_entropy(logits, targets) if targets is not None else None   # compute_loss (sm:91-99)
    return loss, logits


def avclip_forward(sd, vis, aud, logit_scale=0.07, vfeat_all=None, afeat_all=None, chunk=None, drop_path=None):
    """Stage-1 AVCLIP.forward with alpha = 0 (train_clip_src/open_clip/model.py:475-533): towers with
    agg_time_module='AveragePooling' (mean over the t aggregated tokens, motionformer.py:139,246 / ast.py:88), DoNothingBridge
    projections, F.normalize, sim = feat @ feat_all^T / logit_scale, eye(n, m) soft targets (:512-518), symmetric CE (:520-523).
    `*_all` default to the local features (world_size == 1 / gather_for_loss False).  -> dict(vfeat, afeat, sim_v2a, sim_a2v, loss)."""
    vfeat = F.normalize(extract_vfeats(vis, sd, chunk, drop_path=drop_path).mean(2).flatten(0, 1), dim=-1)      # (B*S, D)
    afeat = F.normalize(extract_afeats(aud, sd).mean(2).flatten(0, 1), dim=-1)
    vfeat_all = vfeat if vfeat_all is None else vfeat_all
    afeat_all = afeat if afeat_all is None else afeat_all
    sim_v2a = vfeat @ afeat_all.mT / logit_scale
    sim_a2v = afeat @ vfeat_all.mT / logit_scale
    tgt = torch.eye(*sim_v2a.shape, dtype=sim_v2a.dtype)
    loss = (F.cross_entropy(sim_v2a, tgt) + F.cross_entropy(sim_a2v, tgt)) / 2
    return dict(vfeat=vfeat, afeat=afeat, sim_v2a=sim_v2a, sim_a2v=sim_a2v, loss=loss)


def shift_and_get_preds(a, v, W):
    """Stage-1 zero-shot read-out (train_clip_src/training/train.py:549-579): a, v (B, S, D) -> (preds_a, preds_v) int64 (B, S - W + 1): every
    W-segment window of A against every window of V (window similarity = sum of its W segment dot products), argmax along each axis.
    Pinned to the real function's outputs by tests/golden/shift_preds.npz."""
    B, S, D = a.shape
    n = S - W + 1
    g = torch.einsum('bsd,btd->bst', a.float(), v.float())                                   # per-clip segment similarity
    sim = torch.stack([torch.stack([sum(g[:, i + w, j + w] for w in range(W)) for j in range(n)], -1) for i in range(n)], -2)   # (B, n, n)
    return torch.argmax(sim, dim=-2), torch.argmax(sim, dim=-1)


# ----------------------------------------------------------------------------------------------------
# Deterministic input front-ends (dataset/transforms.py)
# ----------------------------------------------------------------------------------------------------
def rgb_frontend(u8):
    """RGBToHalfToZeroOne -> RGBNormalize(mean .5, std .5) (dataset/transforms.py:647-669; sync.yaml:178-182):
    u8 -> half/255 -> (x-0.5)/0.5.  Returned as fp32 of the fp16-rounded values the reference would feed."""
    x = (u8.half() / 255)
    return ((x - 0.5) / 0.5).float()


def _hz_to_mel_htk(f):
    return 2595.0 * torch.log10(1.0 + f / 700.0)


def mel_filterbank(n_freqs=513, f_min=0.0, f_max=8000.0, n_mels=128, sample_rate=16000):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') restated from its documented
    algorithm (torchaudio is not in this image; checked against an independent per-filter evaluation of the HTK triangles in
    tests/test_oracle_cpu.py::test_mel_frontend_independent_numpy_scipy).  -> (n_freqs, n_mels)."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs, dtype=torch.float64)
    m_pts = torch.linspace(_hz_to_mel_htk(torch.tensor(f_min, dtype=torch.float64)),
                           _hz_to_mel_htk(torch.tensor(f_max, dtype=torch.float64)), n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)             # (n_freqs, n_mels+2)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0).float()


def mel_frontend(wave, pad_to=66, mean=-4.2677393, std=4.5689974):
    """AudioMelSpectrogram -> AudioLog -> PadOrTruncate -> AudioNormalizeAST -> PermuteStreams
    (dataset/transforms.py:815-889; params sync.yaml:183-202): MelSpectrogram(sr 16000, win 400, hop 160,
    n_fft 1024, n_mels 128; torchaudio defaults: periodic Hann zero-padded to n_fft, center+reflect, power 2,
    htk, norm None, f_max sr/2) -> log(x+1e-6) -> right-pad time to 66 with 0 -> (x-mean)/(2*std).
    wave (..., n) fp32 -> (..., 1, 128, 66).  torchaudio is absent from every image of this build, so this is pinned indirectly: torch.stft IS the
    primitive torchaudio's Spectrogram calls with these arguments, and tests/test_oracle_cpu.py::test_mel_frontend_independent_numpy_scipy
    checks the whole chain against a numpy / scipy implementation that shares no code with it (2e-4)."""
    lead = wave.shape[:-1]
    w = wave.reshape(-1, wave.shape[-1]).float()
    win = torch.hann_window(400, periodic=True, dtype=torch.float32)
    spec = torch.stft(w, n_fft=1024, hop_length=160, win_length=400, window=win, center=True,
                      pad_mode='reflect', normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                           # (N, 513, frames)
    mel = torch.matmul(power.transpose(1, 2), mel_filterbank()).transpose(1, 2)   # (N, 128, frames)
    x = torch.log(mel + 1e-6)
    if x.shape[-1] < pad_to:
        x = F.pad(x, (0, pad_to - x.shape[-1]), value=0.0)
    else:
        x = x[..., :pad_to]
    x = (x - mean) / (2 * std)
    return x.reshape(*lead, 1, 128, pad_to)
