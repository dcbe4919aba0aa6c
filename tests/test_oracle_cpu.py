"""CPU: the oracle against (a) golden outputs of the REAL reference and (b) the reference itself when it is importable
(build container only - /root/reference does not exist on the GPU box)."""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

GOLD = Path(__file__).resolve().parent / 'golden'
sys.path.insert(0, str(GOLD))


@pytest.fixture(scope='module')
def sd():
    from synchformer_amd import synth
    return synth.make_state_dict(1337)


def test_oracle_matches_golden_clip0(sd):
    """Full-size forward of clip 0 (14 segments) against the reference's logits + segment features (fp32 vs fp32)."""
    from synchformer_amd import synth
    from oracle import synchformer_cpu as O
    g = np.load(GOLD / 'e2e_sync_B2.npz')
    vis = O.rgb_frontend(synth.make_video_u8(2, 14, 1337))[:1]
    aud = synth.make_spectrogram(2, 14, 1337)[:1]
    with torch.no_grad():
        vf = O.extract_vfeats(vis, sd, chunk=7)
        af = O.extract_afeats(aud, sd)
        v, a = O._lin(vf, sd, 'vproj'), O._lin(af, sd, 'aproj')
        logits = O.global_transformer(v.reshape(1, -1, 768), a.reshape(1, -1, 768), sd)
    gv = torch.from_numpy(g['vfeat_extractor__spatial_attn_agg']).reshape(2, 14, 8, 768)[:1]
    ga = torch.from_numpy(g['afeat_extractor__freq_attn_agg']).reshape(2, 14, 6, 768)[:1]
    assert (vf - gv).abs().max() < 2e-5 and (af - ga).abs().max() < 2e-5
    assert (v - torch.from_numpy(g['vproj'])[:1]).abs().max() < 2e-5
    assert (logits - torch.from_numpy(g['logits'])[:1]).abs().max() < 2e-5
    loss = torch.nn.functional.cross_entropy(logits, torch.from_numpy(g['targets'])[:1])
    assert torch.isfinite(loss)


def test_golden_fixture_integrity():
    for name, keys in [('e2e_sync_B2.npz', ('logits', 'loss', 'vfeat_extractor__blocks__0', 'transformer__ln_f')),
                       ('e2e_sync_gain2_B2.npz', ('logits',)), ('e2e_syncability_B1.npz', ('logits',))]:
        g = np.load(GOLD / name)
        for k in keys:
            assert np.isfinite(g[k]).all(), (name, k)
    g1, g2 = np.load(GOLD / 'e2e_sync_B2.npz'), np.load(GOLD / 'e2e_sync_gain2_B2.npz')
    assert g1['logits'].shape == (2, 21) and np.abs(g2['logits'][0] - g2['logits'][1]).max() > 0.1   # input-sensitive


def test_frontends():
    from oracle import synchformer_cpu as O
    u8 = torch.arange(256, dtype=torch.uint8).reshape(1, 1, 1, 1, 16, 16).expand(1, 1, 1, 3, 16, 16)
    x = O.rgb_frontend(u8)
    assert x.min() == -1.0 and x.max() == 1.0 and x.dtype == torch.float32
    ref = ((u8.half() / 255) - 0.5) / 0.5
    assert torch.equal(x, ref.float())
    wave = torch.sin(torch.arange(10240) * 2 * np.pi * 440 / 16000).reshape(1, 1, -1)
    m = O.mel_frontend(wave)
    assert m.shape == (1, 1, 1, 128, 66) and torch.isfinite(m).all()
    assert torch.all(m[..., 65] == (0.0 + 4.2677393) / (2 * 4.5689974))           # right-padded frame (PadOrTruncate)
    peak = m[0, 0, 0, :, 10].argmax().item()
    fb = O.mel_filterbank()
    assert fb.shape == (513, 128) and abs(fb[:, peak].argmax().item() * 8000 / 512 - 440) < 40   # 440 Hz lands in its band


@pytest.mark.skipif(not (Path('/root/reference/model/sync_model.py').exists()), reason='reference not present (GPU box)')
def test_oracle_matches_real_reference(sd):
    """Import the real reference, load the same synthetic weights, compare on 2 segments + the sync transformer."""
    import ref_import
    from synchformer_amd import synth
    from oracle import synchformer_cpu as O
    m = ref_import.build_reference_synchformer()
    assert list(m.state_dict().keys()) == list(sd.keys())
    m.load_state_dict(sd, strict=True)
    vis = O.rgb_frontend(synth.make_video_u8(1, 2, 7))
    aud = synth.make_spectrogram(1, 2, 7)
    g = torch.Generator().manual_seed(1)
    v, a = torch.randn(2, 112, 768, generator=g), torch.randn(2, 84, 768, generator=g)
    with torch.no_grad():
        assert (m.extract_vfeats(vis, for_loop=False) - O.extract_vfeats(vis, sd)).abs().max() < 1e-5
        assert (m.extract_afeats(aud, for_loop=False) - O.extract_afeats(aud, sd)).abs().max() < 1e-5
        assert (m.transformer(v, a) - O.global_transformer(v, a, sd)).abs().max() < 1e-5


def test_oracle_avclip_matches_golden_towers():
    """Stage-1 towers (AveragePooling) of the REAL reference + the contrastive head (fixture: tests/golden/make_golden.py avclip)."""
    from synchformer_amd import synth
    from oracle import synchformer_cpu as O
    g = np.load(GOLD / 'avclip_towers_B2S3.npz')
    B, S = int(g['B']), int(g['S'])
    sd2 = synth.make_state_dict(1337, gain=float(g['gain']))
    vis = O.rgb_frontend(synth.make_video_u8(B, S, 1337))
    aud = synth.make_spectrogram(B, S, 1337)
    with torch.no_grad():
        out = O.avclip_forward(sd2, vis, aud, logit_scale=float(g['logit_scale']))
        vseg = O.extract_vfeats(vis, sd2).mean(2)
    assert (vseg - torch.from_numpy(g['ref_vseg'])).abs().max() < 2e-5
    rv = torch.nn.functional.normalize(torch.from_numpy(g['ref_vseg']).flatten(0, 1), dim=-1)
    ra = torch.nn.functional.normalize(torch.from_numpy(g['ref_aseg']).flatten(0, 1), dim=-1)
    assert (out['vfeat'] - rv).abs().max() < 1e-5 and (out['afeat'] - ra).abs().max() < 1e-5
    assert (out['sim_v2a'] - torch.from_numpy(g['restated_sim_v2a'])).abs().max() < 2e-4
    assert abs(float(out['loss']) - float(g['restated_loss'])) < 1e-4


def test_oracle_token_masks_match_golden():
    """vis_mask / aud_mask (NaN-trick token masks, key masking in every attention) against the REAL reference (make_golden.py masked)."""
    from synchformer_amd import synth
    from oracle import synchformer_cpu as O
    g = np.load(GOLD / 'e2e_masked_B1S2.npz')
    B, S = int(g['B']), int(g['S'])
    sdm = synth.make_state_dict(1337, gain=float(g['gain']), n_pos=2 + S * 14)
    vis = O.rgb_frontend(synth.make_video_u8(B, S, 1337))
    aud = synth.make_spectrogram(B, S, 1337)
    vm, am = synth.make_masks(B, S, 1337)
    with torch.no_grad():
        vf = O.extract_vfeats(vis, sdm, vis_mask=vm)
        af = O.extract_afeats(aud, sdm, aud_mask=am)
        _, logits = O.synchformer_forward(sdm, vis, aud, vis_mask=vm, aud_mask=am)
    assert (vf.reshape(-1, 768) - torch.from_numpy(g['vfeat']).reshape(-1, 768)).abs().max() < 5e-5
    assert (af.reshape(-1, 768) - torch.from_numpy(g['afeat']).reshape(-1, 768)).abs().max() < 5e-5
    assert (logits - torch.from_numpy(g['logits'])).abs().max() < 5e-5
    assert (torch.from_numpy(g['logits']) - torch.from_numpy(g['logits_nomask'])).abs().max() > 0.1     # the masks matter
