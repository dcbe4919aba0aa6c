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


def test_mel_frontend_independent_numpy_scipy():
    """An independent pin of the mel front-end (dataset/transforms.py:815-889): torchaudio is absent, so `oracle.mel_frontend` (torch.stft -
    the primitive torchaudio.transforms.MelSpectrogram itself calls - + the closed-form HTK filterbank) is checked against a second
    implementation that shares no code with it: explicit reflect padding and framing in numpy, scipy.signal.get_window('hann') (periodic),
    np.fft.rfft, and the HTK triangles evaluated per filter from the textbook definition (linear ramps between mel-equidistant corner
    frequencies, hz = 700 (10^(m / 2595) - 1)).  Known answers of the HTK scale ride along."""
    from scipy.signal import get_window
    from oracle import synchformer_cpu as O
    rng = np.random.default_rng(7)
    n = 10240                                                        # one 0.64 s segment
    wave = (0.3 * rng.standard_normal((3, n)) + np.sin(np.arange(n) * 2 * np.pi * 1000 / 16000)).astype(np.float32)
    n_fft, win_len, hop, n_mels, sr = 1024, 400, 160, 128, 16000
    win = np.zeros(n_fft)
    win[(n_fft - win_len) // 2:(n_fft - win_len) // 2 + win_len] = get_window('hann', win_len, fftbins=True)   # centred in the FFT frame
    padded = np.pad(wave.astype(np.float64), ((0, 0), (n_fft // 2, n_fft // 2)), mode='reflect')
    n_frames = 1 + n // hop
    frames = np.stack([padded[:, t * hop:t * hop + n_fft] for t in range(n_frames)], 1) * win       # (3, frames, n_fft)
    power = np.abs(np.fft.rfft(frames, axis=-1)) ** 2                                                # (3, frames, 513)
    mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    imel = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    assert abs(mel(1000.0) - 999.9855) < 1e-3 and abs(mel(8000.0) - 2840.0230) < 1e-3                 # HTK known answers
    corners = imel(np.linspace(mel(0.0), mel(sr / 2), n_mels + 2))
    freqs = np.arange(n_fft // 2 + 1) * sr / n_fft
    fb = np.zeros((n_fft // 2 + 1, n_mels))
    for j in range(n_mels):
        lo, c, hi = corners[j], corners[j + 1], corners[j + 2]
        up, down = (freqs - lo) / (c - lo), (hi - freqs) / (hi - c)
        fb[:, j] = np.clip(np.minimum(up, down), 0.0, None)
    x = np.log(power @ fb + 1e-6).transpose(0, 2, 1)                                                  # (3, 128, frames)
    out = np.zeros((3, n_mels, 66))
    out[:, :, :min(66, n_frames)] = x[:, :, :66]
    out = (out - (-4.2677393)) / (2 * 4.5689974)
    got = O.mel_frontend(torch.from_numpy(wave)).reshape(3, n_mels, 66).numpy()
    assert n_frames == 65 and np.abs(got - out).max() < 2e-4, np.abs(got - out).max()
    assert np.abs(O.mel_filterbank().numpy() - fb).max() < 1e-6


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
    # head against the REAL AVCLIP class (forward / forward_for_logging / compute_loss on gathered features)
    assert (out['vfeat'] - torch.from_numpy(g['ref_vfeat'])).abs().max() < 1e-5
    assert (out['sim_v2a'] - torch.from_numpy(g['ref_sim_v2a'])).abs().max() < 2e-4
    assert (out['sim_a2v'] - torch.from_numpy(g['ref_sim_a2v'])).abs().max() < 2e-4
    assert abs(float(out['loss']) - float(g['ref_loss'])) < 1e-4
    rvf, raf = torch.from_numpy(g['ref_vfeat']), torch.from_numpy(g['ref_afeat'])
    for r in range(2):                          # two ranks of one clip each, features gathered in rank order: eye(n, m) targets on every rank
        loc = slice(r * S, (r + 1) * S)
        sim = rvf[loc] @ raf.mT / 0.07
        sim2 = raf[loc] @ rvf.mT / 0.07
        tgt = torch.eye(*sim.shape)
        loss = (torch.nn.functional.cross_entropy(sim, tgt) + torch.nn.functional.cross_entropy(sim2, tgt)) / 2
        assert abs(float(loss) - float(g[f'gathered_loss_rank{r}'])) < 1e-5
    with torch.no_grad():
        out_lo = O.avclip_forward(sd2, vis, aud, logit_scale=float(g['clamped_lo']))
    assert abs(float(out_lo['loss']) - float(g['loss_at_clamped_lo'])) < 1e-2 * float(g['loss_at_clamped_lo'])


def test_oracle_shift_preds_match_reference():
    """oracle.shift_and_get_preds against the REAL function's outputs (tests/golden/shift_preds.npz)."""
    from oracle import synchformer_cpu as O
    g = np.load(GOLD / 'shift_preds.npz')
    for tag in 'abcd':
        pa, pv = O.shift_and_get_preds(torch.from_numpy(g[f'{tag}_a']), torch.from_numpy(g[f'{tag}_v']), int(g[f'{tag}_W']))
        assert torch.equal(pa, torch.from_numpy(g[f'{tag}_preds_a'])) and torch.equal(pv, torch.from_numpy(g[f'{tag}_preds_v']))


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
