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
