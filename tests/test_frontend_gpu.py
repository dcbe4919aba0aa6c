"""GPU: the device mel front-end (sf_mel_frontend) against the oracle's torch.stft-based restatement.  NOTE: that
restatement follows torchaudio's documented algorithm but torchaudio itself is not available -> parity unpinned."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('kind', ['uniform', 'tones', 'short'])
def test_mel_frontend_matches_oracle(gpu, kind):
    from synchformer_amd import synth
    from synchformer_amd.frontend import MelFrontend
    from oracle import synchformer_cpu as O
    if kind == 'uniform':
        wave = synth.make_wave(2, 14, 1337)                                   # (2, 14, 10240) U(-1, 1)
    elif kind == 'tones':
        t = torch.arange(10240) / 16000.0
        wave = torch.stack([torch.sin(2 * torch.pi * f * t) * a for f, a in [(100., .5), (440., 1.), (3000., .1), (7900., .7)]]).reshape(1, 4, -1)
    else:
        wave = synth.make_wave(1, 3, 5)[..., :5000]                            # fewer frames than pad_to -> more padding
    ref = O.mel_frontend(wave)
    got = MelFrontend(gpu)(wave.to(gpu)).cpu()
    assert got.shape == ref.shape == (*wave.shape[:-1], 1, 128, 66)
    # normalised log-mel: 1e-3 here = 1e-2 in log units.  Broadband input: fp32 DFT-by-matmul vs torch's FFT agree to 2e-4.
    # Pure tones: bins far from the tone hold leakage ~1e-7 of the peak, i.e. fp32 cancellation noise next to the 1e-6 floor
    # inside the log, where the two (equally valid) fp32 summation orders differ by up to ~1e-2 in log units.
    torch.testing.assert_close(got, ref, rtol=0, atol=2e-3 if kind == 'tones' else 2e-4)


def test_mel_filterbank_host_table_matches_oracle():
    from synchformer_amd.frontend import mel_filterbank
    from oracle import synchformer_cpu as O
    torch.testing.assert_close(torch.from_numpy(mel_filterbank()), O.mel_filterbank(), rtol=1e-6, atol=1e-7)


def test_device_side_segmenting_matches_host_slicing(gpu):
    """forward_clips (segments read in place from the clip by sf_im2col_video_clips / sf_mel_frontend_clips) must equal the
    reference's order of operations: slice overlapping segments on the host (GenerateMultipleSegments), then forward."""
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    from synchformer_amd.frontend import MelFrontend, segment_ranges
    eng = SynchformerEngine(synth.make_state_dict(1337, gain=2.0), gpu, seg_chunk=14)
    mel = MelFrontend(gpu)
    B, T, n_samp = 3, 125, 80000
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (B, T, 3, 224, 224), generator=g, dtype=torch.uint8)
    wave = synth.make_wave(B, 1, 5, n=n_samp).reshape(B, n_samp)
    r = segment_ranges(T, n_samp)
    vis = torch.stack([torch.stack([frames[b, r['v_start'] + s * 8: r['v_start'] + s * 8 + 16] for s in range(14)]) for b in range(B)])
    aw = torch.stack([torch.stack([wave[b, r['a_start'] + s * 5120: r['a_start'] + s * 5120 + 10240] for s in range(14)]) for b in range(B)])
    ref_aud = mel(aw.to(gpu))                                                  # (B, 14, 1, 128, 66)
    got_aud = mel.segments(wave.to(gpu), r['a_start'], r['a_stride'], 14, r['a_size'])
    assert torch.equal(ref_aud, got_aud)
    ref = eng.forward(vis.to(gpu), ref_aud)
    got = eng.forward_clips(frames.to(gpu), wave.to(gpu), mel)
    assert torch.equal(ref, got)
    with pytest.raises(RuntimeError, match='do not fit'):
        mel.segments(wave[:, :70000].to(gpu), r['a_start'], r['a_stride'], 14, r['a_size'])


def test_host_clip_pipeline_equals_direct_forward(gpu):
    """HostClipPipeline (pinned host buffers -> copy stream -> two device slots -> forward_clips) returns, batch after batch, exactly what forward_clips returns
    on the same clips uploaded by hand - including when a slot is reused (the third batch overwrites the first one's slot only after its forward)."""
    from synchformer_amd import synth
    from synchformer_amd.engine import SynchformerEngine
    from synchformer_amd.frontend import HostClipPipeline, MelFrontend
    eng = SynchformerEngine(synth.make_state_dict(1337, gain=2.0), gpu, seg_chunk=28)
    mel = MelFrontend(gpu)
    B, T, n_samp = 2, 125, 80000
    g = torch.Generator().manual_seed(11)
    batches = [HostClipPipeline.pinned_like(torch.randint(0, 256, (B, T, 3, 224, 224), generator=g, dtype=torch.uint8),
                                            torch.rand(B, n_samp, generator=g) * 2 - 1) for _ in range(4)]
    want = [eng.forward_clips(f.to(gpu), w.to(gpu), mel).clone() for f, w in batches]
    assert not torch.equal(want[0], want[1])
    pipe = HostClipPipeline(eng, mel, B, T, n_samp)
    pipe.stage(*batches[0])
    got = []
    for i in range(4):
        nxt = batches[i + 1] if i + 1 < 4 else (None, None)
        got.append(pipe.step(*nxt).clone())
    torch.cuda.synchronize()
    for i in range(4):
        assert torch.equal(got[i], want[i]), i
    with pytest.raises(RuntimeError, match='nothing staged'):
        pipe.step()
    pipe.stage(*batches[0]); pipe.stage(*batches[1])
    with pytest.raises(RuntimeError, match='both slots'):
        pipe.stage(*batches[2])
    with pytest.raises(ValueError, match='expected uint8'):
        HostClipPipeline(eng, mel, B, T, n_samp).stage(batches[0][0][:1], batches[0][1])
    with pytest.raises(ValueError, match='expected uint8'):                                  # a non-fp32 wave would take a converting (synchronous) copy
        HostClipPipeline(eng, mel, B, T, n_samp).stage(batches[0][0], batches[0][1].double())
    # the recycling contract: stage() hands back the event behind its copies; after wait_staged() the host buffers may be overwritten
    p2 = HostClipPipeline(eng, mel, B, T, n_samp)
    f_host, w_host = batches[3][0].clone().pin_memory(), batches[3][1].clone().pin_memory()
    ev = p2.stage(f_host, w_host)
    p2.wait_staged()
    assert ev.query()
    f_host.zero_(); w_host.zero_()
    assert torch.equal(p2.step(), want[3])
