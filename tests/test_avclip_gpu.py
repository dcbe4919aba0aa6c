"""Stage-1 (segment-level contrastive, AVCLIP) forward on the GPU through the drop-in class: towers with AveragePooling, F.normalize,
fp32 similarity and symmetric cross-entropy, against (a) real-reference tower outputs (tests/golden/avclip_towers_B2S3.npz) and
(b) torch fp32 restatements of the small head kernels.  Tolerances: pooled tower features relative RMS <= 1.5 % (bf16 GEMM
operands, same bar as tests/test_e2e_gpu.py); cosine similarities |delta| <= 2e-3, i.e. 3e-2 after the 1/0.07 temperature."""
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / 'golden'


def _rel_rms(a, b):
    return ((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()


@pytest.mark.parametrize('n,t', [(1, 1), (5, 8), (28, 6), (224, 8)])
@pytest.mark.parametrize('normalize', [False, True])
def test_meanpool_l2norm(gpu, n, t, normalize):
    from synchformer_amd import ops
    x = torch.randn(n * t, 768, device=gpu)
    if normalize and n > 1:
        x[t:2 * t] = 0                                             # zero vector: F.normalize's eps clamp
    out = ops.meanpool_l2norm(x, torch.empty(n, 768, device=gpu), t, normalize)
    ref = x.view(n, t, 768).double().mean(1)
    if normalize:
        ref = torch.nn.functional.normalize(ref, dim=-1)
    assert (out.double() - ref).abs().max().item() < 2e-6


@pytest.mark.parametrize('n,m,d', [(1, 1, 16), (6, 6, 768), (28, 224, 768), (70, 130, 64), (224, 224, 768)])
def test_similarity_f32(gpu, n, m, d):
    from synchformer_amd import ops
    a, b = torch.randn(n, d, device=gpu), torch.randn(m, d, device=gpu)
    out = ops.similarity(a, b, torch.full((n, m), float('nan'), device=gpu), 1 / 0.07)
    ref = (a.double() @ b.double().T) / 0.07
    assert (out.double() - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item() / 100)


def _model(gpu, gain, gather=False):
    import synchformer_amd as sa
    from synchformer_amd import synth
    m = sa.instantiate_from_config(sa.avclip_yaml_model_config(gather_for_loss=gather))
    sd = synth.make_state_dict(1337, gain=gain)
    own = {k.replace('vfeat_extractor.', 'v_encoder.').replace('afeat_extractor.', 'a_encoder.'): v for k, v in sd.items()
           if k.startswith(('vfeat_extractor.', 'afeat_extractor.'))}
    own['logit_scale'] = torch.tensor(0.07)
    m.load_state_dict(own, strict=True)
    return m.to(gpu).eval()


def test_avclip_forward_matches_reference_towers(gpu):
    from synchformer_amd import synth
    from oracle import synchformer_cpu as O
    g = np.load(GOLD / 'avclip_towers_B2S3.npz')
    B, S, gain = int(g['B']), int(g['S']), float(g['gain'])
    m = _model(gpu, gain)
    assert len(m.state_dict()) == 451
    vis = O.rgb_frontend(synth.make_video_u8(B, S, 1337)).permute(0, 1, 3, 2, 4, 5).to(gpu)     # (B, S, C, Tv, H, W)
    aud = synth.make_spectrogram(B, S, 1337).squeeze(2).permute(0, 1, 3, 2).contiguous().to(gpu)  # (B, S, Ta, F)
    with pytest.raises(NotImplementedError, match='no backward'):
        m.encode_streams(vis, aud)                              # feature extraction with autograd on: loud, not silently detached
    with torch.no_grad():
        out = m(vis, aud)
        log = m.forward_for_logging(vis, aud)
        vseg, _ = m.v_encoder(vis)
        aseg, _ = m.a_encoder(aud)
    ev, ea = _rel_rms(vseg.cpu(), torch.from_numpy(g['ref_vseg'])), _rel_rms(aseg.cpu(), torch.from_numpy(g['ref_aseg']))
    vfeat, afeat = out['rgb_features'][0].cpu(), out['audio_features'][0].cpu()
    assert vfeat.shape == (B * S, 768) and out['rgb_features'][1] is None and out['logit_scales'][1] is None
    assert (vfeat.norm(dim=-1) - 1).abs().max() < 1e-5
    # the REAL AVCLIP class' outputs (forward / forward_for_logging of open_clip/model.py:475-567; cosines = sim * logit_scale)
    cos_ref = torch.from_numpy(g['ref_sim_v2a']) * float(g['logit_scale'])
    dcos = (log['segment_sim_v2a'].cpu() * 0.07 - cos_ref).abs().max().item()
    dloss = abs(float(out['losses']['segment_contrastive_loss']) - float(g['ref_loss']))
    print(f'avclip: vseg relrms {ev:.4f} aseg relrms {ea:.4f} | cos max {dcos:.5f} | loss delta {dloss:.5f}')
    assert ev < 1.5e-2 and ea < 1.5e-2
    assert dcos < 2e-3 and dloss < 1e-2
    for k in ('v2a', 'a2v', 'v2v', 'a2a'):
        d = (log[f'segment_sim_{k}'].cpu() - torch.from_numpy(g[f'ref_sim_{k}'])).abs().max().item() * 0.07
        assert d < 2e-3, (k, d)
    assert (vfeat - torch.from_numpy(g['ref_vfeat'])).abs().max() < 2e-3 and (afeat - torch.from_numpy(g['ref_afeat'])).abs().max() < 2e-3
    assert abs(float(log['segment_contrastive_loss']) - float(g['ref_logging_loss'])) < 1e-2
    # gathered features (gather_for_loss, world 2): the reference's eye(n, m) targets put the positive of LOCAL row i at column i on every rank
    # (open_clip/model.py:489-512); HIP head fed the reference's own features so that only the convention is under test
    eng = m._engine()
    rv, ra = torch.from_numpy(g['ref_vfeat']).to(gpu), torch.from_numpy(g['ref_afeat']).to(gpu)
    for r in range(2):
        loc = slice(r * S, (r + 1) * S)
        losses, sim_v2a, _ = eng.contrastive_loss(rv[loc].contiguous(), ra[loc].contiguous(), rv, ra, 0.07)
        assert abs(float(losses.mean()) - float(g[f'gathered_loss_rank{r}'])) < 2e-4, r
        assert (sim_v2a.cpu() - torch.from_numpy(g[f'gathered_sim_v2a_rank{r}'])).abs().max() < 2e-4
    assert torch.allclose(log['segment_sim_a2v'], log['segment_sim_v2a'].T, atol=1e-5)
    assert abs(float(log['segment_contrastive_loss']) - float(out['losses']['segment_contrastive_loss'])) < 1e-6
    # head kernels alone, fed the HIP features: exact fp32 restatement
    sim = vfeat @ afeat.T / 0.07
    tgt = torch.eye(B * S)
    loss = (torch.nn.functional.cross_entropy(sim, tgt) + torch.nn.functional.cross_entropy(sim.T, tgt)) / 2
    assert (log['segment_sim_v2a'].cpu() - sim).abs().max() < 1e-4
    assert abs(float(loss) - float(out['losses']['segment_contrastive_loss'])) < 1e-5
    # clamp_logit_scales (open_clip/model.py:569-572)
    with torch.no_grad():
        m.logit_scale.fill_(0.9)
        assert abs(float(m(vis, aud)['logit_scales'][0]) - float(g['clamped_hi'])) < 1e-7
        m.logit_scale.fill_(1e-5)
        lo = m(vis, aud)
    assert abs(float(lo['logit_scales'][0]) - float(g['clamped_lo'])) < 1e-9
    # temperature 0.001: cosines are amplified 1000x, so the loss tolerance scales with the cosine error (2e-3 -> ~2 in logits); the bar is relative
    assert abs(float(lo['losses']['segment_contrastive_loss']) - float(g['loss_at_clamped_lo'])) < 0.05 * float(g['loss_at_clamped_lo']) + 0.5


@pytest.mark.parametrize('B,S,W', [(1, 14, 8), (3, 14, 8), (2, 9, 1), (2, 6, 5)])
def test_shift_and_get_preds(gpu, B, S, W):
    """Zero-shot shifted-window predictions against the reference's unfold + matmul + argmax formulation (training/train.py:549-579)."""
    from synchformer_amd.stage1 import shift_and_get_preds
    torch.manual_seed(B * 100 + S)
    a = torch.nn.functional.normalize(torch.randn(B, S, 768), dim=-1)
    v = torch.nn.functional.normalize(a + 0.8 * torch.randn(B, S, 768), dim=-1)
    pa, pv = shift_and_get_preds(a.to(gpu), v.to(gpu), W)
    af = a.unfold(-2, W, 1).contiguous().view(B, S - W + 1, -1)
    vf = v.unfold(-2, W, 1).contiguous().view(B, S - W + 1, -1)
    sim = af.double() @ vf.double().mT
    assert torch.equal(pa.cpu(), torch.argmax(sim, dim=-2)) and torch.equal(pv.cpu(), torch.argmax(sim, dim=-1))


@pytest.mark.parametrize('tag', ['a', 'b', 'c', 'd'])
def test_shift_and_get_preds_reference_golden(gpu, tag):
    """Same read-out against the REAL `shift_and_get_preds` (train_clip_src/training/train.py:549-579) run on seeded features
    (tests/golden/shift_preds.npz, made by make_golden.py shift_preds).  Integer outputs: exact."""
    from synchformer_amd.stage1 import shift_and_get_preds
    g = np.load(GOLD / 'shift_preds.npz')
    pa, pv = shift_and_get_preds(torch.from_numpy(g[f'{tag}_a']).to(gpu), torch.from_numpy(g[f'{tag}_v']).to(gpu), int(g[f'{tag}_W']))
    assert torch.equal(pa.cpu(), torch.from_numpy(g[f'{tag}_preds_a'])) and torch.equal(pv.cpu(), torch.from_numpy(g[f'{tag}_preds_v']))


def test_eval_one_example(gpu):
    from synchformer_amd import synth
    from synchformer_amd.stage1 import eval_one_example
    from oracle import synchformer_cpu as O
    m = _model(gpu, 2.0)
    B, S = 1, 10
    vis = O.rgb_frontend(synth.make_video_u8(B, S, 1337)).permute(0, 1, 3, 2, 4, 5).to(gpu)
    aud = synth.make_spectrogram(B, S, 1337).squeeze(2).permute(0, 1, 3, 2).contiguous().to(gpu)
    losses, metrics = eval_one_example(m, vis, aud, win=8)
    assert torch.isfinite(losses['segment_contrastive_loss']) and 0.0 <= float(metrics['precision']) <= 1.0
    assert abs(float(metrics['precision']) - (float(metrics['precision_a']) + float(metrics['precision_v'])) / 2) < 1e-6
