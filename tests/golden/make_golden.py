"""Generate golden vectors by running the REAL reference (/root/reference) on CPU in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Inputs and weights are regenerated from seeds by `synchformer_amd.synth` (numpy Philox, bit-reproducible),
so the fixtures hold only seeds + expected OUTPUTS (data, not reference source).  The reference cannot
travel to the GPU box; these files can.
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent.parent))

import ref_import  # noqa: E402
from synchformer_amd import synth  # noqa: E402

SEED = 1337
TOK_V = [0, 1, 2, 197, 1000, 1568]   # token rows of the (1569) visual sequence kept in fixtures
TOK_A = [0, 1, 2, 3, 40, 73]         # token rows of the (74) audio sequence kept in fixtures


def rgb_frontend_ref(u8):
    # the reference's own two transforms (dataset/transforms.py:653, :657-669 -> torchvision Normalize),
    # applied verbatim in half precision; torchvision is absent so Normalize is spelled out.
    x = u8.half().div(255.)
    mean = torch.as_tensor([0.5, 0.5, 0.5], dtype=x.dtype).view(-1, 1, 1)
    std = torch.as_tensor([0.5, 0.5, 0.5], dtype=x.dtype).view(-1, 1, 1)
    return x.sub(mean).div(std)


def capture(model, names):
    store, hooks = {}, []
    mods = dict(model.named_modules())

    def mk(n):
        def hook(_m, _inp, out):
            o = out[0] if isinstance(out, tuple) else out
            if hasattr(o, 'last_hidden_state'):
                o = o.last_hidden_state
            store[n] = o.detach().float()
        return hook
    for n in names:
        hooks.append(mods[n].register_forward_hook(mk(n)))
    return store, hooks


def e2e_sync(B=2, gain=1.0):
    """gain=1: reference-like init scale.  gain=2: sharper attention / stronger input dependence (logit spread between
    clips ~0.2 vs ~0.02), so an input-blind bug cannot hide inside the tolerance."""
    model = ref_import.build_reference_synchformer()
    sd = synth.make_state_dict(SEED, gain=gain)
    model.load_state_dict(sd, strict=True)
    vis = rgb_frontend_ref(synth.make_video_u8(B, 14, SEED)).float()
    aud = synth.make_spectrogram(B, 14, SEED)
    tgt = synth.make_targets(B, 21, SEED)
    v, a = 'vfeat_extractor', 'afeat_extractor'
    names = [f'{v}.patch_embed_3d', f'{v}.blocks.0.timeattn', f'{v}.blocks.0.attn', f'{v}.blocks.0',
             f'{v}.blocks.5', f'{v}.blocks.11', f'{v}.norm', f'{v}.spatial_attn_agg',
             f'{a}.ast.embeddings', f'{a}.ast.encoder.layer.0', f'{a}.ast.encoder.layer.11', f'{a}.ast.layernorm',
             f'{a}.freq_attn_agg', 'vproj', 'aproj', 'transformer.blocks.0', 'transformer.ln_f']
    store, hooks = capture(model, names)
    with torch.no_grad():
        loss, logits = model(vis, aud, tgt)
    for h in hooks:
        h.remove()
    out = dict(seed=np.int64(SEED), B=np.int64(B), gain=np.float64(gain), logits=logits.numpy(), loss=loss.numpy(), targets=tgt.numpy())
    for n, t in store.items():
        key = n.replace('.', '__')
        if n.startswith(v) and t.dim() == 3 and t.shape[1] in (1568, 1569):
            rows = [r for r in TOK_V if r < t.shape[1]]
            out[key] = t[[0, 13, -1]][:, rows].numpy()          # segments 0, 13 and last
        elif n.startswith(a) and t.dim() == 3 and t.shape[1] == 74:
            out[key] = t[[0, 13, -1]][:, TOK_A].numpy()
        else:
            out[key] = t.reshape(-1, t.shape[-1])[:64].numpy() if t.numel() > 200000 else t.numpy()
        print(n, tuple(t.shape), '->', out[key].shape)
    tag = '' if gain == 1.0 else f'_gain{gain:g}'
    np.savez_compressed(HERE / f'e2e_sync{tag}_B{B}.npz', **out)
    print('logits', logits, 'loss', loss)


def e2e_syncability(B=1):
    """configs/ft_synchability.yaml: S=13 segments, 184-token pos_emb, 2-way sync_head (sync_model.py:176-190)."""
    model = ref_import.build_reference_synchformer(
        n_segments_tokens=184, transformer_target='model.sync_model.GlobalTransformerWithSyncabilityHead')
    sd = synth.make_state_dict(SEED, n_pos=184, n_out=2, head='sync_head')
    ref_keys = list(model.state_dict().keys())
    assert ref_keys == list(sd.keys()), set(ref_keys) ^ set(sd.keys())
    model.load_state_dict(sd, strict=True)
    vis = rgb_frontend_ref(synth.make_video_u8(B, 13, SEED)).float()
    aud = synth.make_spectrogram(B, 13, SEED)
    with torch.no_grad():
        _, logits = model(vis, aud)
    np.savez_compressed(HERE / f'e2e_syncability_B{B}.npz', seed=np.int64(SEED), B=np.int64(B), logits=logits.numpy())
    print('syncability logits', logits)


def train_grads(B=2):
    """Stage-2 backward (train_utils.py:199-204: extractors frozen) of the REAL reference: CE loss + gradients of vproj, aproj and
    the sync transformer, fed with the golden segment features, dropout disabled (eval()), fp32."""
    g = np.load(HERE / f'e2e_sync_B{B}.npz')
    model = ref_import.build_reference_synchformer()
    model.load_state_dict(synth.make_state_dict(SEED), strict=True)
    model.eval()
    vf = torch.from_numpy(g['vfeat_extractor__spatial_attn_agg']).reshape(B, 14, 8, 768)
    af = torch.from_numpy(g['afeat_extractor__freq_attn_agg']).reshape(B, 14, 6, 768)
    tgt = torch.from_numpy(g['targets'])
    train = [(n, p) for n, p in model.named_parameters() if n.startswith(('vproj.', 'aproj.', 'transformer.'))]
    for p in model.parameters():
        p.requires_grad_(False)
    for _, p in train:
        p.requires_grad_(True)
    v, a = model.vproj(vf), model.aproj(af)
    logits = model.transformer(v.view(B, -1, 768), a.view(B, -1, 768))
    loss = model.compute_loss(logits, tgt)
    loss.backward()
    out = dict(loss=loss.detach().numpy(), logits=logits.detach().numpy())
    keep = ('transformer.off_head.bias', 'transformer.off_head.weight', 'transformer.ln_f.weight', 'transformer.OFF_tok', 'transformer.MOD_tok',
            'vproj.bias', 'aproj.bias', 'transformer.blocks.0.attn.query.bias', 'transformer.blocks.2.mlp.2.bias',
            'transformer.blocks.1.ln1.weight', 'transformer.vis_in_lnorm.weight')
    names, norms = [], []
    for n, p in train:
        names.append(n); norms.append(float(p.grad.norm()))
        if n in keep:
            out['grad__' + n.replace('.', '__')] = p.grad.numpy()
    out['names'] = np.array(names); out['grad_norms'] = np.array(norms, dtype=np.float64)
    out['grad__transformer__blocks__0__mlp__0__weight__rows0_4'] = dict(train)['transformer.blocks.0.mlp.0.weight'].grad[:4].numpy()
    out['grad__transformer__pos_emb__rows0_4'] = dict(train)['transformer.pos_emb_cfg.pos_emb'].grad[0, :4].numpy()
    np.savez_compressed(HERE / f'train_sync_B{B}_grads.npz', **out)
    print('train loss', float(loss), 'total grad norm', float(np.sqrt((np.array(norms) ** 2).sum())))


def avclip(B=2, S=3, gain=2.0):
    """Stage-1 towers (configs/segment_avclip.yaml: agg_time_module 'AveragePooling') run through the REAL MotionFormer / AST.
    The AVCLIP class itself cannot be imported here (its package __init__ needs torchvision, absent from this image), so the
    5-line head (F.normalize, sim / scale, eye targets, symmetric CE; open_clip/model.py:506-533) is applied to the real
    tower outputs with plain torch below - the fixture marks which arrays are reference outputs and which are restated."""
    ref = ref_import.import_reference()
    tower = dict(ckpt_path=None, extract_features=True, agg_time_module='AveragePooling', add_global_repr=False,
                 agg_segments_module='AveragePooling', max_segments=14)
    with ref_import._cwd(ref_import.REF):
        vt = ref['MotionFormer'](factorize_space_time=True, agg_space_module='TransformerEncoderLayer', **tower).eval()
        at = ref['AST'](max_spec_t=66, factorize_freq_time=True, agg_freq_module='TransformerEncoderLayer', **tower).eval()
    sd = synth.make_state_dict(SEED, gain=gain)
    vsd = {k[len('vfeat_extractor.'):]: v for k, v in sd.items() if k.startswith('vfeat_extractor.')}
    asd = {k[len('afeat_extractor.'):]: v for k, v in sd.items() if k.startswith('afeat_extractor.')}
    assert list(vt.state_dict().keys()) == list(vsd.keys()) and list(at.state_dict().keys()) == list(asd.keys())
    vt.load_state_dict(vsd, strict=True)
    at.load_state_dict(asd, strict=True)
    vis = rgb_frontend_ref(synth.make_video_u8(B, S, SEED)).float()           # (B, S, Tv, C, H, W)
    aud = synth.make_spectrogram(B, S, SEED)                                   # (B, S, 1, F, Ta)
    with torch.no_grad():
        vseg, _ = vt(vis.permute(0, 1, 3, 2, 4, 5), False)                     # AVCLIP feeds (B, S, C, Tv, H, W)
        aseg, _ = at(aud.squeeze(2).permute(0, 1, 3, 2), False)                # and (B, S, Ta, F)
        vfeat = torch.nn.functional.normalize(vseg.flatten(0, 1), dim=-1)
        afeat = torch.nn.functional.normalize(aseg.flatten(0, 1), dim=-1)
        scale = 0.07
        sim_v2a, sim_a2v = vfeat @ afeat.mT / scale, afeat @ vfeat.mT / scale
        tgt = torch.eye(*sim_v2a.shape)
        loss = (torch.nn.functional.cross_entropy(sim_v2a, tgt) + torch.nn.functional.cross_entropy(sim_a2v, tgt)) / 2
    np.savez_compressed(HERE / f'avclip_towers_B{B}S{S}.npz', seed=np.int64(SEED), B=np.int64(B), S=np.int64(S), gain=np.float64(gain),
                        logit_scale=np.float64(scale), ref_vseg=vseg.numpy(), ref_aseg=aseg.numpy(),
                        restated_sim_v2a=sim_v2a.numpy(), restated_sim_a2v=sim_a2v.numpy(), restated_loss=loss.numpy())
    print('avclip towers', tuple(vseg.shape), tuple(aseg.shape), 'loss', float(loss), 'sim spread', float(sim_v2a.max() - sim_v2a.min()))


def _real_towers(gain):
    ref = ref_import.import_reference()
    tower = dict(ckpt_path=None, extract_features=True, agg_time_module='AveragePooling', add_global_repr=False,
                 agg_segments_module='AveragePooling', max_segments=14)
    with ref_import._cwd(ref_import.REF):
        vt = ref['MotionFormer'](factorize_space_time=True, agg_space_module='TransformerEncoderLayer', **tower).eval()
        at = ref['AST'](max_spec_t=66, factorize_freq_time=True, agg_freq_module='TransformerEncoderLayer', **tower).eval()
    sd = synth.make_state_dict(SEED, gain=gain)
    vt.load_state_dict({k[len('vfeat_extractor.'):]: v for k, v in sd.items() if k.startswith('vfeat_extractor.')}, strict=True)
    at.load_state_dict({k[len('afeat_extractor.'):]: v for k, v in sd.items() if k.startswith('afeat_extractor.')}, strict=True)
    return vt, at


def avclip_grads(B=1, S=3, gain=2.0):
    """Stage-1 backward through the REAL MotionFormer / AST towers (fp32 autograd, eval mode = no DropPath), with the AVCLIP head
    restated as in `avclip` above: loss + per-parameter gradient norms + a few full gradients."""
    vt, at = _real_towers(gain)
    vis = rgb_frontend_ref(synth.make_video_u8(B, S, SEED)).float()
    aud = synth.make_spectrogram(B, S, SEED)
    scale = torch.tensor(0.07, requires_grad=True)
    vseg, _ = vt(vis.permute(0, 1, 3, 2, 4, 5), False)
    aseg, _ = at(aud.squeeze(2).permute(0, 1, 3, 2), False)
    vfeat = torch.nn.functional.normalize(vseg.flatten(0, 1), dim=-1)
    afeat = torch.nn.functional.normalize(aseg.flatten(0, 1), dim=-1)
    sim_v2a, sim_a2v = vfeat @ afeat.mT / scale, afeat @ vfeat.mT / scale
    tgt = torch.eye(*sim_v2a.shape)
    loss = (torch.nn.functional.cross_entropy(sim_v2a, tgt) + torch.nn.functional.cross_entropy(sim_a2v, tgt)) / 2
    loss.backward()
    named = [('vfeat_extractor.' + n, p) for n, p in vt.named_parameters()] + [('afeat_extractor.' + n, p) for n, p in at.named_parameters()]
    named = [(n, p) for n, p in named if p.grad is not None] + [('logit_scale', scale)]
    keep = ('vfeat_extractor.cls_token', 'vfeat_extractor.temp_embed', 'vfeat_extractor.blocks.0.timeattn.qkv.bias', 'vfeat_extractor.blocks.0.norm3.weight',
            'vfeat_extractor.blocks.11.attn.proj.bias', 'vfeat_extractor.blocks.5.mlp.fc2.bias', 'vfeat_extractor.norm.weight',
            'vfeat_extractor.spatial_attn_agg.cls_token', 'vfeat_extractor.spatial_attn_agg.self_attn.in_proj_bias',
            'vfeat_extractor.patch_embed_3d.proj.bias', 'afeat_extractor.ast.embeddings.cls_token', 'afeat_extractor.ast.embeddings.position_embeddings',
            'afeat_extractor.ast.encoder.layer.0.attention.attention.query.bias', 'afeat_extractor.ast.encoder.layer.11.output.dense.bias',
            'afeat_extractor.ast.layernorm.weight', 'afeat_extractor.freq_attn_agg.linear1.bias', 'afeat_extractor.ast.embeddings.patch_embeddings.projection.bias')
    out = dict(seed=np.int64(SEED), B=np.int64(B), S=np.int64(S), gain=np.float64(gain), loss=loss.detach().numpy(), logit_scale_grad=scale.grad.numpy())
    names, norms = [], []
    for n, p in named:
        names.append(n); norms.append(float(p.grad.norm()))
        if n in keep:
            out['grad__' + n.replace('.', '__')] = p.grad.numpy()
    out['names'] = np.array(names); out['grad_norms'] = np.array(norms, dtype=np.float64)
    d = dict(named)
    out['gradrows__vfeat_extractor__blocks__0__timeattn__qkv__weight'] = d['vfeat_extractor.blocks.0.timeattn.qkv.weight'].grad[[0, 768, 1536, 2303]].numpy()
    out['gradrows__vfeat_extractor__blocks__6__attn__qkv__weight'] = d['vfeat_extractor.blocks.6.attn.qkv.weight'].grad[[0, 768, 1536, 2303]].numpy()
    out['gradrows__vfeat_extractor__blocks__11__mlp__fc1__weight'] = d['vfeat_extractor.blocks.11.mlp.fc1.weight'].grad[[0, 1000, 2000, 3071]].numpy()
    out['gradrows__afeat_extractor__ast__encoder__layer__3__attention__attention__key__weight'] = \
        d['afeat_extractor.ast.encoder.layer.3.attention.attention.key.weight'].grad[[0, 768 // 2, 767]].numpy()
    np.savez_compressed(HERE / f'avclip_grads_B{B}S{S}.npz', **out)
    print('avclip grads: loss', float(loss), 'total grad norm', float(np.sqrt((np.array(norms) ** 2).sum())), 'n tensors', len(names),
          'dscale', float(scale.grad))


def e2e_masked(B=1, S=2, gain=2.0):
    """Synchformer.forward with vis_mask / aud_mask (sync_model.py:38-89; token masks via the NaN trick) through the REAL reference."""
    n_pos = 2 + S * 14
    model = ref_import.build_reference_synchformer(n_segments_tokens=n_pos)
    model.load_state_dict(synth.make_state_dict(SEED, gain=gain, n_pos=n_pos), strict=True)
    vis = rgb_frontend_ref(synth.make_video_u8(B, S, SEED)).float()
    aud = synth.make_spectrogram(B, S, SEED)
    vm, am = synth.make_masks(B, S, SEED)
    names = ['vfeat_extractor.spatial_attn_agg', 'afeat_extractor.freq_attn_agg', 'vfeat_extractor.blocks.0', 'afeat_extractor.ast.encoder.layer.0']
    store, hooks = capture(model, names)
    with torch.no_grad():
        _, logits = model(vis, aud, vis_mask=vm, aud_mask=am)
        _, logits_nomask = model(vis, aud)
    for h in hooks:
        h.remove()
    out = dict(seed=np.int64(SEED), B=np.int64(B), S=np.int64(S), gain=np.float64(gain), logits=logits.numpy(), logits_nomask=logits_nomask.numpy())
    # hooks fired twice: `store` holds the UNMASKED pass; rerun for the masked one
    store, hooks = capture(model, names)
    with torch.no_grad():
        model(vis, aud, vis_mask=vm, aud_mask=am)
    for h in hooks:
        h.remove()
    out['vfeat'] = store['vfeat_extractor.spatial_attn_agg'].numpy()
    out['afeat'] = store['afeat_extractor.freq_attn_agg'].numpy()
    out['vblock0_rows'] = store['vfeat_extractor.blocks.0'][:, TOK_V].numpy()
    out['ablock0_rows'] = store['afeat_extractor.ast.encoder.layer.0'][:, TOK_A].numpy()
    np.savez_compressed(HERE / f'e2e_masked_B{B}S{S}.npz', **out)
    print('masked logits', logits, 'unmasked', logits_nomask, 'max diff', float((logits - logits_nomask).abs().max()))


if __name__ == '__main__':
    torch.manual_seed(0)
    which = sys.argv[1:] or ['sync', 'sync_gain2', 'syncability', 'train', 'avclip', 'avclip_grads', 'masked']
    if 'sync' in which:
        e2e_sync(2)
    if 'sync_gain2' in which:
        e2e_sync(2, gain=2.0)
    if 'syncability' in which:
        e2e_syncability(1)
    if 'train' in which:
        train_grads(2)
    if 'avclip' in which:
        avclip(2, 3)
    if 'avclip_grads' in which:
        avclip_grads(1, 3)
    if 'masked' in which:
        e2e_masked(1, 2)
