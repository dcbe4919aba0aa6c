"""Generate golden vectors by running the REAL reference (/root/reference) on CPU in the build container.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Inputs and weights are regenerated from seeds by `synchformer_amd.synth` (numpy Philox, bit-reproducible),
so the fixtures hold only seeds + expected OUTPUTS (data, not reference source).  The reference cannot
travel to the GPU box; these files can.
"""
import os
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


def train_grads_ft(B=2, S=13):
    """Synchronizability fine-tune step (configs/ft_synchability.yaml: frozen extractors, GlobalTransformerWithSyncabilityHead over 13 segments =
    184 tokens, 2-way sync_head on token 0, sync_model.py:176-190) through the REAL reference: features of both extractors, logits, CE loss and
    the gradients of vproj / aproj / the transformer, eval mode (dropout off), fp32."""
    model = ref_import.build_reference_synchformer(
        n_segments_tokens=184, transformer_target='model.sync_model.GlobalTransformerWithSyncabilityHead')
    sd = synth.make_state_dict(SEED, n_pos=184, n_out=2, head='sync_head')
    model.load_state_dict(sd, strict=True)
    model.eval()
    vis = rgb_frontend_ref(synth.make_video_u8(B, S, SEED)).float()
    aud = synth.make_spectrogram(B, S, SEED)
    tgt = torch.tensor([1, 0][:B], dtype=torch.int64)
    with torch.no_grad():
        vf = model.extract_vfeats(vis, for_loop=False)
        af = model.extract_afeats(aud, for_loop=False)
    train = [(n, p_) for n, p_ in model.named_parameters() if n.startswith(('vproj.', 'aproj.', 'transformer.'))]
    for p_ in model.parameters():
        p_.requires_grad_(False)
    for _, p_ in train:
        p_.requires_grad_(True)
    v, a = model.vproj(vf), model.aproj(af)
    logits = model.transformer(v.view(B, -1, 768), a.view(B, -1, 768))
    loss = model.compute_loss(logits, tgt)
    loss.backward()
    out = dict(seed=np.int64(SEED), B=np.int64(B), S=np.int64(S), loss=loss.detach().numpy(), logits=logits.detach().numpy(), targets=tgt.numpy(),
               vfeat=vf.numpy(), afeat=af.numpy())
    keep = ('transformer.sync_head.bias', 'transformer.sync_head.weight', 'transformer.ln_f.weight', 'transformer.OFF_tok', 'transformer.MOD_tok',
            'vproj.bias', 'aproj.bias', 'transformer.blocks.0.attn.query.bias', 'transformer.blocks.2.mlp.2.bias', 'transformer.vis_in_lnorm.weight')
    names, norms = [], []
    for n, p_ in train:
        names.append(n); norms.append(float(p_.grad.norm()))
        if n in keep:
            out['grad__' + n.replace('.', '__')] = p_.grad.numpy()
    out['names'] = np.array(names); out['grad_norms'] = np.array(norms, dtype=np.float64)
    out['grad__transformer__pos_emb__rows0_4'] = dict(train)['transformer.pos_emb_cfg.pos_emb'].grad[0, :4].numpy()
    np.savez_compressed(HERE / f'train_ft_B{B}_grads.npz', **out)
    print('ft train: logits', logits.detach(), 'loss', float(loss), 'total grad norm', float(np.sqrt((np.array(norms) ** 2).sum())), 'n tensors', len(names))


def _real_avclip(gain, gather_for_loss=False):
    """The REAL AVCLIP (train_clip_src/open_clip/model.py:449-585) built from configs/segment_avclip.yaml's model section (interpolations
    resolved, ckpt_path null) with the synthetic weights loaded into its two towers."""
    import synchformer_amd as sa
    ref = ref_import.import_reference_avclip()
    cfg = sa.avclip_yaml_model_config(gather_for_loss=gather_for_loss)['params']
    with ref_import._cwd(ref_import.REF):
        m = ref['AVCLIP'](**cfg).eval()
    sd = synth.make_state_dict(SEED, gain=gain)
    own = {'v_encoder.' + k[len('vfeat_extractor.'):]: v for k, v in sd.items() if k.startswith('vfeat_extractor.')}
    own.update({'a_encoder.' + k[len('afeat_extractor.'):]: v for k, v in sd.items() if k.startswith('afeat_extractor.')})
    own['logit_scale'] = m.logit_scale.detach().clone()
    assert set(own) == set(m.state_dict()), set(own) ^ set(m.state_dict())
    m.load_state_dict(own, strict=True)
    return m


def avclip(B=2, S=3, gain=2.0):
    """Stage-1 `AVCLIP` (configs/segment_avclip.yaml) - the REAL class: forward() (loss with local features), forward_for_logging() (the four
    similarity matrices + loss), and compute_loss() fed with features 'gathered' from two ranks, which pins the reference's eye(n, m) target
    convention under gather_for_loss (open_clip/model.py:489-512: the positive of local row i is column i on EVERY rank)."""
    m = _real_avclip(gain)
    vis = rgb_frontend_ref(synth.make_video_u8(B, S, SEED)).float()           # (B, S, Tv, C, H, W)
    aud = synth.make_spectrogram(B, S, SEED)                                   # (B, S, 1, F, Ta)
    v_in, a_in = vis.permute(0, 1, 3, 2, 4, 5), aud.squeeze(2).permute(0, 1, 3, 2)   # AVCLIP is fed (B, S, C, Tv, H, W) / (B, S, Ta, F)
    with torch.no_grad():
        vseg, _ = m.v_encoder(v_in, False)
        aseg, _ = m.a_encoder(a_in, False)
        out = m(v_in, a_in)
        log = m.forward_for_logging(v_in, a_in)
        vfeat, afeat = out['rgb_features'][0], out['audio_features'][0]
        # two "ranks" of one clip each (n = S rows per rank, m = 2 S gathered columns): what forward() computes per rank when
        # world_size == 2 and gather_for_loss (torch.distributed.nn.all_gather concatenates in rank order)
        n = S
        gath = {}
        for r in range(2):
            loss_r, (s_v2a, s_a2v) = m.compute_loss(vfeat[r * n:(r + 1) * n], afeat[r * n:(r + 1) * n], vfeat.mT, afeat.mT, m.logit_scale, alpha=0)
            gath[f'gathered_loss_rank{r}'] = loss_r.numpy()
            gath[f'gathered_sim_v2a_rank{r}'] = s_v2a.numpy()
        # clamp_logit_scales (open_clip/model.py:569-572): a scale outside [clamp_scale_min, clamp_scale_max] is clamped in place by forward()
        m.logit_scale.data.fill_(0.9)
        clamped_hi = float(m(v_in, a_in)['logit_scales'][0])
        m.logit_scale.data.fill_(1e-5)
        out_lo = m(v_in, a_in)
    np.savez_compressed(HERE / f'avclip_towers_B{B}S{S}.npz', seed=np.int64(SEED), B=np.int64(B), S=np.int64(S), gain=np.float64(gain),
                        logit_scale=np.float64(0.07), ref_vseg=vseg.numpy(), ref_aseg=aseg.numpy(),
                        ref_vfeat=vfeat.numpy(), ref_afeat=afeat.numpy(), ref_loss=out['losses']['segment_contrastive_loss'].numpy(),
                        ref_sim_v2a=log['segment_sim_v2a'].numpy(), ref_sim_a2v=log['segment_sim_a2v'].numpy(),
                        ref_sim_v2v=log['segment_sim_v2v'].numpy(), ref_sim_a2a=log['segment_sim_a2a'].numpy(),
                        ref_logging_loss=log['segment_contrastive_loss'].numpy(), clamped_hi=np.float64(clamped_hi),
                        clamped_lo=np.float64(float(out_lo['logit_scales'][0])), loss_at_clamped_lo=out_lo['losses']['segment_contrastive_loss'].numpy(),
                        **gath)
    print('avclip (real class)', tuple(vseg.shape), tuple(aseg.shape), 'loss', float(out['losses']['segment_contrastive_loss']),
          'gathered losses', float(gath['gathered_loss_rank0']), float(gath['gathered_loss_rank1']), 'clamps', clamped_hi, float(out_lo['logit_scales'][0]))


def shift_preds():
    """Stage-1 zero-shot read-out `shift_and_get_preds` (train_clip_src/training/train.py:549-579) - the REAL function (its module imports
    torchvision / torchaudio / matplotlib at module scope: import-only stand-ins, tests/golden/ref_shims_late) on seeded random features."""
    ref_import.import_reference_avclip()
    with ref_import._cwd(ref_import.REF):
        from model.modules.feat_extractors.train_clip_src.training.train import shift_and_get_preds
    out = {}
    for tag, (B, S, D, W) in {'a': (3, 14, 768, 8), 'b': (2, 14, 768, 14), 'c': (4, 9, 64, 1), 'd': (1, 14, 768, 5)}.items():
        g = torch.Generator().manual_seed(100 + B * S + W)
        a = torch.nn.functional.normalize(torch.randn(B, S, D, generator=g), dim=-1)
        v = torch.nn.functional.normalize(a + (10.0 if D > 64 else 0.8) * torch.randn(B, S, D, generator=g), dim=-1)   # weakly correlated: a non-trivial argmax
        pa, pv = shift_and_get_preds(a, v, W)
        out[f'{tag}_a'], out[f'{tag}_v'] = a.numpy(), v.numpy()
        out[f'{tag}_W'] = np.int64(W)
        out[f'{tag}_preds_a'], out[f'{tag}_preds_v'] = pa.numpy(), pv.numpy()
        print('shift_preds', tag, (B, S, D, W), pa.tolist(), pv.tolist())
    np.savez_compressed(HERE / 'shift_preds.npz', **out)


def segments():
    """GenerateMultipleSegments (dataset/transforms.py:400-500; is_start_random False, no jitter) - the REAL transform's segment ranges for a
    grid of clip lengths / segment counts / strides.  Pins synchformer_amd.frontend.segment_ranges (device-side segmenting)."""
    ref_import.import_reference_avclip()                                       # late shims: the module imports torchvision / torchaudio
    with ref_import._cwd(ref_import.REF):
        from dataset.transforms import GenerateMultipleSegments
    cases = []
    for v_len, a_len, v_fps, a_fps, seg_v, n_seg, step in [
            (125, 80000, 25, 16000, 16, 14, 0.5),        # configs/sync.yaml: 5 s crop, 14 half-overlapping segments
            (250, 160000, 25, 16000, 16, 14, 0.5), (250, 160000, 25, 16000, 16, 0, 0.5), (125, 80000, 25, 16000, 16, 0, 1.0),
            (120, 76800, 25, 16000, 16, 13, 0.5),       # configs/ft_synchability.yaml: 13 segments
            (131, 83210, 25, 16000, 16, 14, 0.5), (200, 127999, 25, 16000, 16, 0, 0.5), (125, 110250, 25, 22050, 16, 14, 0.5),
            (150, 80000, 30, 16000, 16, 12, 0.5), (125, 80000, 25, 16000, 8, 20, 0.75), (124, 79360, 25, 16000, 16, 14, 0.5),
            (100, 64000, 25, 16000, 16, 14, 0.5), (125, 60000, 25, 16000, 16, 14, 0.5)]:   # too short for 14 segments (video / audio): the transform asserts
        t = GenerateMultipleSegments(segment_size_vframes=seg_v, n_segments=n_seg or None, is_start_random=False, step_size_seg=step)
        item = dict(video=torch.zeros(v_len, 1, 1, 1), audio=torch.arange(a_len, dtype=torch.float32), path='synthetic',
                    meta=dict(video=dict(fps=[v_fps]), audio=dict(framerate=[a_fps])))
        try:
            out = t(item)
            a0 = out['audio'][:, 0].long().tolist()                           # audio sample index = its value
            nseg, a_size = out['audio'].shape
            v_size = out['video'].shape[1]
            cases.append([v_len, a_len, v_fps, a_fps, seg_v, n_seg, int(round(step * 100)), 1, nseg, v_size, a_size] + a0 + [-1] * (32 - nseg))
        except AssertionError:
            cases.append([v_len, a_len, v_fps, a_fps, seg_v, n_seg, int(round(step * 100)), 0, 0, 0, 0] + [-1] * 32)
        print('segments', cases[-1][:11], cases[-1][11:11 + max(cases[-1][8], 0)][:4], '...')
    # the video starts are recovered the same way from a frame-index video
    vcases = []
    for c in cases:
        v_len, a_len, v_fps, a_fps, seg_v, n_seg, step100, ok = c[:8]
        if not ok:
            vcases.append([-1] * 32)
            continue
        t = GenerateMultipleSegments(segment_size_vframes=seg_v, n_segments=n_seg or None, is_start_random=False, step_size_seg=step100 / 100)
        item = dict(video=torch.arange(v_len, dtype=torch.float32).view(v_len, 1, 1, 1), audio=torch.zeros(a_len), path='synthetic',
                    meta=dict(video=dict(fps=[v_fps]), audio=dict(framerate=[a_fps])))
        v0 = t(item)['video'][:, 0, 0, 0, 0].long().tolist()
        vcases.append(v0 + [-1] * (32 - len(v0)))
    np.savez_compressed(HERE / 'segment_ranges.npz', cases=np.array(cases, dtype=np.int64), v_starts=np.array(vcases, dtype=np.int64),
                        columns=np.array(['v_len', 'a_len', 'v_fps', 'a_fps', 'seg_v', 'n_seg(0=max)', 'step*100', 'ok', 'n_out', 'v_size', 'a_size',
                                          'a_start[0..31]']))


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


def avclip_grads_full(B=2, S=14, gain=2.0, chunk=2):
    """Stage 1 at its CONFIGURED geometry (configs/segment_avclip.yaml:61 base_batch_size 2 x 14 segments = a 28 x 28 contrastive problem)
    through the REAL AVCLIP class, eval mode (no DropPath), fp32.  The towers treat segments independently, so the backward is run `chunk`
    segments at a time from the exact feature gradients of the real compute_loss (identical to one big backward, a fraction of the memory).
    Stored: loss, the similarity matrix, d loss / d logit_scale, and the gradient NORM of every parameter tensor (+ a few small gradients)."""
    m = _real_avclip(gain)
    vis = rgb_frontend_ref(synth.make_video_u8(B, S, SEED)).float().permute(0, 1, 3, 2, 4, 5)       # (B, S, C, Tv, H, W)
    aud = synth.make_spectrogram(B, S, SEED).squeeze(2).permute(0, 1, 3, 2)                          # (B, S, Ta, F)
    n = B * S
    v_flat, a_flat = vis.reshape(1, n, *vis.shape[2:]), aud.reshape(1, n, *aud.shape[2:])
    with torch.no_grad():
        vseg = torch.cat([m.v_encoder(v_flat[:, i:i + chunk], False)[0] for i in range(0, n, chunk)], 1)[0]      # (n, 768)
        aseg = torch.cat([m.a_encoder(a_flat[:, i:i + chunk], False)[0] for i in range(0, n, chunk)], 1)[0]
    vl, al = vseg.clone().requires_grad_(True), aseg.clone().requires_grad_(True)
    vf, af = torch.nn.functional.normalize(vl, dim=-1), torch.nn.functional.normalize(al, dim=-1)
    loss, (sim_v2a, _) = m.compute_loss(vf, af, vf.mT, af.mT, m.logit_scale, alpha=0)
    loss.backward()
    dscale = m.logit_scale.grad.clone()
    for i in range(0, n, chunk):
        vo = m.v_encoder(v_flat[:, i:i + chunk], False)[0][0]
        vo.backward(vl.grad[i:i + chunk])
        ao = m.a_encoder(a_flat[:, i:i + chunk], False)[0][0]
        ao.backward(al.grad[i:i + chunk])
        print('avclip_grads_full: segments', i, '..', i + chunk, flush=True)
    named = [('vfeat_extractor.' + k, p_) for k, p_ in m.v_encoder.named_parameters()] + [('afeat_extractor.' + k, p_) for k, p_ in m.a_encoder.named_parameters()]
    named = [(k, p_) for k, p_ in named if p_.grad is not None]
    keep = ('vfeat_extractor.cls_token', 'vfeat_extractor.temp_embed', 'vfeat_extractor.blocks.0.norm3.weight', 'vfeat_extractor.blocks.11.attn.proj.bias',
            'vfeat_extractor.norm.weight', 'vfeat_extractor.spatial_attn_agg.cls_token', 'afeat_extractor.ast.embeddings.cls_token',
            'afeat_extractor.ast.encoder.layer.11.output.dense.bias', 'afeat_extractor.freq_attn_agg.linear1.bias')
    out = dict(seed=np.int64(SEED), B=np.int64(B), S=np.int64(S), gain=np.float64(gain), loss=loss.detach().numpy(), logit_scale_grad=dscale.numpy(),
               sim_v2a=sim_v2a.detach().numpy(), names=np.array([k for k, _ in named] + ['logit_scale']),
               grad_norms=np.array([float(p_.grad.norm()) for _, p_ in named] + [float(dscale.abs())], dtype=np.float64))
    for k, p_ in named:
        if k in keep:
            out['grad__' + k.replace('.', '__')] = p_.grad.numpy()
    np.savez_compressed(HERE / f'avclip_grads_B{B}S{S}.npz', **out)
    print('avclip_grads_full: loss', float(loss), 'dscale', float(dscale), 'total grad norm', float(np.sqrt((out['grad_norms'] ** 2).sum())),
          'sim spread', float(sim_v2a.max() - sim_v2a.min()))


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


def e2e_masked_fused(B=1, S=6, gain=2.0, seed=77):
    """token masks at a batch large enough for the FUSED launches (6 segments = 9,414 token rows >= engine.py's 128 * 64 threshold:
    sf_qkv_space_attention_masked / sf_qkv_time_attention2_masked / sf_gemm_res_ln768) through the REAL reference (sync_model.py:72-89, vit_helper.py:107-141),
    with one whole frame, one whole wave, one whole patch column and the left-over patches masked (synth.make_masks_fused_case)."""
    n_pos = 2 + S * 14
    model = ref_import.build_reference_synchformer(n_segments_tokens=n_pos)
    model.load_state_dict(synth.make_state_dict(SEED, gain=gain, n_pos=n_pos), strict=True)
    vis = rgb_frontend_ref(synth.make_video_u8(B, S, seed)).float()
    aud = synth.make_spectrogram(B, S, seed)
    vm, am = synth.make_masks_fused_case(B, S, seed)
    names = ['vfeat_extractor.spatial_attn_agg', 'afeat_extractor.freq_attn_agg', 'vfeat_extractor.blocks.0', 'vfeat_extractor.blocks.11']
    with torch.no_grad():
        _, logits_nomask = model(vis, aud)
    store, hooks = capture(model, names)
    with torch.no_grad():
        _, logits = model(vis, aud, vis_mask=vm, aud_mask=am)
    for h in hooks:
        h.remove()
    out = dict(seed=np.int64(seed), B=np.int64(B), S=np.int64(S), gain=np.float64(gain), logits=logits.numpy(), logits_nomask=logits_nomask.numpy(),
               vfeat=store['vfeat_extractor.spatial_attn_agg'].numpy(), afeat=store['afeat_extractor.freq_attn_agg'].numpy(),
               vblock0_rows=store['vfeat_extractor.blocks.0'][:, TOK_V].numpy(), vblock11_rows=store['vfeat_extractor.blocks.11'][:, TOK_V].numpy())
    np.savez_compressed(HERE / f'e2e_masked_B{B}S{S}.npz', **out)
    print('masked (fused-schedule case) logits', logits, 'unmasked', logits_nomask, 'max diff', float((logits - logits_nomask).abs().max()))


SYNC_HEAD_SCALE = 5.0     # 'trained' variant of the synchronizability fixture: sync_head.weight x this on top of gain-2 weights


def syncability_state_dict(variant):
    """The two inits of the syncability_logits fixture (configs/ft_synchability.yaml: 184-token pos_emb, 2-way sync_head): 'gain1' = the reference-like init;
    'trained' = gain-2 weights with the head scaled so that l1 - l0 has a trained model's spread.  tests/test_e2e_gpu.py rebuilds the same dicts on the GPU box."""
    if variant == 'gain1':
        return synth.make_state_dict(SEED, n_pos=184, n_out=2, head='sync_head')
    sd = synth.make_state_dict(SEED, gain=2.0, n_pos=184, n_out=2, head='sync_head')
    sd['transformer.sync_head.weight'] = sd['transformer.sync_head.weight'] * SYNC_HEAD_SCALE
    return sd


def syncability_logits(n_clips=32, per=2, variants=('gain1', 'trained')):
    """VERDICT r5 item 3b - BASELINE configs[4]'s Acc@1-parity number: `n_clips` structured 13-segment clips through the REAL reference with the
    GlobalTransformerWithSyncabilityHead (sync_model.py:176-190), 2-way logits only, at the reference-like init and at a trained scale.  The GPU test asserts the
    2-way argmax agreement and max |d(l1 - l0)| for the bf16 engine AND for the MXFP8 towers configs[4] runs on."""
    out = dict(seed=np.int64(SEED), n_clips=np.int64(n_clips), head_scale=np.float64(SYNC_HEAD_SCALE))
    part = Path(os.environ.get('SYNCABILITY_PARTIAL', '/tmp/syncability_logits_partial.npz'))          # resumable
    done = dict(np.load(part)) if part.exists() else {}
    for variant in variants:
        model = None
        rows = []
        for c0 in range(0, n_clips, per):
            key = f'{variant}_{c0}'
            if key not in done:
                if model is None:
                    model = ref_import.build_reference_synchformer(n_segments_tokens=184, transformer_target='model.sync_model.GlobalTransformerWithSyncabilityHead')
                    model.load_state_dict(syncability_state_dict(variant), strict=True)
                    model.eval()
                u8, aud = synth.make_structured_clips(c0, min(per, n_clips - c0), 13, SEED)
                with torch.no_grad():
                    _, logits = model(rgb_frontend_ref(u8).float(), aud)
                done[key] = logits.numpy()
                np.savez(part, **done)
            rows.append(done[key])
            print(variant, c0, rows[-1].tolist(), flush=True)
        out['logits_' + variant] = np.concatenate(rows, 0)
    import zlib
    for c in (0, n_clips - 1):
        u8, aud = synth.make_structured_clip(c, 13, SEED)
        out[f'crc_vis_{c}'] = np.int64(zlib.crc32(u8.numpy().tobytes()))
        out[f'crc_aud_{c}'] = np.int64(zlib.crc32(aud.numpy().tobytes()))
    np.savez_compressed(HERE / f'syncability_logits_{n_clips}.npz', **out)


LOGITS_HEAD_SCALE = 5.0   # 'trained' variant: off_head.weight x this on top of gain-2 weights -> top logit ~ 10 like the released model's (README.md:79)


def logits_state_dict(variant):
    """The two inits of the logits_only fixture: 'gain1' = the reference-like init scale of the other fixtures; 'trained' = gain-2 weights with the offset
    head scaled so that the logits have the scale of a TRAINED model (top logit ~ 10-12, README.md:79) and clips of different content land on
    different classes with trained-size margins.  tests/test_e2e_gpu.py rebuilds the same dict on the GPU box."""
    if variant == 'gain1':
        return synth.make_state_dict(SEED)
    sd = synth.make_state_dict(SEED, gain=2.0)
    sd['transformer.off_head.weight'] = sd['transformer.off_head.weight'] * LOGITS_HEAD_SCALE
    return sd


def logits_only(n_clips=32, per=2, variants=('gain1', 'trained')):
    """Acc@1-parity proxy at scale: `n_clips` STRUCTURED clips (synth.make_structured_clips: clips differ in content) through the REAL
    reference, logits only (n_clips x 21 floats per variant), at the reference-like init AND at a trained-scale init.  The GPU test asserts max |dlogit|,
    100 % argmax agreement and the +-1-class agreement of calc_cls_metrics (scripts/train_utils.py:632)."""
    out = dict(seed=np.int64(SEED), n_clips=np.int64(n_clips), head_scale=np.float64(LOGITS_HEAD_SCALE))
    part = Path(os.environ.get('LOGITS_PARTIAL', '/tmp/logits_only_partial.npz'))          # resumable: ~1.3 min of CPU per reference forward
    done = dict(np.load(part)) if part.exists() else {}
    for variant in variants:
        model = None
        rows = []
        for c0 in range(0, n_clips, per):
            key = f'{variant}_{c0}'
            if key not in done:
                if model is None:
                    model = ref_import.build_reference_synchformer()
                    model.load_state_dict(logits_state_dict(variant), strict=True)
                    model.eval()
                u8, aud = synth.make_structured_clips(c0, min(per, n_clips - c0), 14, SEED)
                with torch.no_grad():
                    _, logits = model(rgb_frontend_ref(u8).float(), aud)
                done[key] = logits.numpy()
                np.savez(part, **done)
            rows.append(done[key])
            print(variant, c0, rows[-1].argmax(1).tolist(), rows[-1].max(1).tolist(), flush=True)
        out['logits_' + variant] = np.concatenate(rows, 0)
    import zlib
    for c in (0, n_clips - 1):                                                # input pins: the GPU box must regenerate exactly these clips
        u8, aud = synth.make_structured_clip(c, 14, SEED)
        out[f'crc_vis_{c}'] = np.int64(zlib.crc32(u8.numpy().tobytes()))
        out[f'crc_aud_{c}'] = np.int64(zlib.crc32(aud.numpy().tobytes()))
    np.savez_compressed(HERE / f'logits_only_{n_clips}.npz', **out)


def checkpoints():
    """what the REAL reference constructors hold after reading the synthetic checkpoint files of ckpt_fixtures.py -
    `MotionFormer(ckpt_path='...epoch_best.pt')` (motionformer.py:52-80, 156-173), `AST(ckpt_path='...epoch_best.pt')` (ast.py:58-60, 113-131) and
    `AST(ckpt_path='MIT/ast-finetuned-audioset-10-10-0.4593')` with its 1214 -> 74 position cut (ast.py:49-53, 240-245).  For the last one the two
    `from_pretrained` calls - the hub download - are pointed at the local synthetic weights; everything behind them is the reference's own code.
    Stored per tensor: sum, sum |x|, first element (float64) + the whole truncated position table."""
    import tempfile
    import ckpt_fixtures as cf
    ref = ref_import.import_reference()
    tower = dict(extract_features=True, agg_time_module='torch.nn.Identity', add_global_repr=False)
    out = {}
    with tempfile.TemporaryDirectory() as td:
        td = Path(td)
        s1 = cf.write_stage1_ckpt(td / 'epoch_best.pt')
        with ref_import._cwd(ref_import.REF):
            vt = ref['MotionFormer'](ckpt_path=str(s1), factorize_space_time=True, agg_space_module='TransformerEncoderLayer', **tower)
            at = ref['AST'](ckpt_path=str(s1), max_spec_t=66, factorize_freq_time=True, agg_freq_module='TransformerEncoderLayer', **tower)
        for tag, m in (('s1_v', vt), ('s1_a', at)):
            names, vals = cf.digest(m.state_dict())
            out[f'{tag}_names'], out[f'{tag}_vals'] = np.array(names), vals
        out['s1_v_patch_embed_requires_grad'] = np.array([p.requires_grad for p in vt.patch_embed.parameters()])
        # HF AST: the hub fetch replaced by the local synthetic weights, the constructor untouched
        import model.modules.feat_extractors.audio.ast as ast_mod
        hf_sd = cf.hf_ast_state()

        def local_model(cls, name, revision=None, **kw):
            assert name == 'MIT/ast-finetuned-audioset-10-10-0.4593'
            cfg = ast_mod.ASTConfig()
            cfg.num_labels = 527
            m = cls(cfg)
            st = m.load_state_dict(hf_sd, strict=True)
            return m

        def local_cfg(cls, name, revision=None, **kw):
            cfg = cls()
            cfg.num_labels = 527
            return cfg
        old_m, old_c = ast_mod.ASTForAudioClassification.from_pretrained, ast_mod.ASTConfig.from_pretrained
        ast_mod.ASTForAudioClassification.from_pretrained = classmethod(local_model)
        ast_mod.ASTConfig.from_pretrained = classmethod(local_cfg)
        try:
            with ref_import._cwd(ref_import.REF):
                ah = ref['AST'](ckpt_path='MIT/ast-finetuned-audioset-10-10-0.4593', max_spec_t=66, factorize_freq_time=True,
                                agg_freq_module='TransformerEncoderLayer', **tower)
        finally:
            ast_mod.ASTForAudioClassification.from_pretrained, ast_mod.ASTConfig.from_pretrained = old_m, old_c
        ast_only = {k: v for k, v in ah.state_dict().items() if k.startswith('ast.')}           # (the aggregator is random-initialised there)
        names, vals = cf.digest(ast_only)
        out['hf_names'], out['hf_vals'] = np.array(names), vals
        out['hf_position_embeddings'] = ah.state_dict()['ast.embeddings.position_embeddings'].numpy()
    np.savez_compressed(HERE / 'checkpoints.npz', **out)
    print('checkpoints:', {k: getattr(v, 'shape', None) for k, v in out.items()})


if __name__ == '__main__':
    torch.manual_seed(0)
    which = sys.argv[1:] or ['sync', 'sync_gain2', 'syncability', 'train', 'train_ft', 'avclip', 'shift_preds', 'segments', 'avclip_grads', 'masked', 'checkpoints']
    if 'sync' in which:
        e2e_sync(2)
    if 'sync_gain2' in which:
        e2e_sync(2, gain=2.0)
    if 'syncability' in which:
        e2e_syncability(1)
    if 'train' in which:
        train_grads(2)
    if 'train_ft' in which:
        train_grads_ft(2)
    if 'avclip' in which:
        avclip(2, 3)
    if 'shift_preds' in which:
        shift_preds()
    if 'segments' in which:
        segments()
    if 'avclip_grads' in which:
        avclip_grads(1, 3)
    if 'avclip_grads_full' in which:
        avclip_grads_full(2, 14)
    if 'masked' in which:
        e2e_masked(1, 2)
    if 'checkpoints' in which:
        checkpoints()
    if 'masked_fused' in which:                                           # ~1 min of CPU; not in the default list
        e2e_masked_fused(1, 6)
    if 'syncability_logits' in which:                                     # ~40 min of CPU (64 reference forwards of 13 segments); not in the default list
        syncability_logits(int(os.environ.get('N_CLIPS', '32')))
    if 'logits_only' in which:                                            # ~25 min of CPU (64 reference forwards); not in the default list
        logits_only(int(os.environ.get('N_CLIPS', '32')))
