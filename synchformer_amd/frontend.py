"""Device front-ends in front of Synchformer.forward.

`MelFrontend` turns 16 kHz waveform segments into the normalised log-mel tensors the model consumes, on the GPU, via
`sf_mel_frontend` (two fp32 kernels).  The tables it uploads - window-folded DFT twiddles and the HTK mel filterbank -
are constants of the transform (torchaudio's documented MelSpectrogram defaults, dataset/transforms.py:815-823 with
configs/sync.yaml:183-188), computed once here in float64.  The RGB front-end (a27) needs no module: uint8 frames are
normalised inside `sf_im2col_video`.
"""
import math

import numpy as np
import torch

from . import _lib

AST_MEAN, AST_STD = -4.2677393, 4.5689974        # configs/sync.yaml:196-197 (AudioNormalizeAST)


def mel_filterbank(n_freqs=513, f_min=0.0, f_max=8000.0, n_mels=128, sample_rate=16000) -> np.ndarray:
    """HTK-scale triangular filterbank without area normalisation, (n_freqs, n_mels) float32."""
    freqs = np.linspace(0.0, sample_rate // 2, n_freqs)
    to_mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    pts = 700.0 * (10.0 ** (np.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2) / 2595.0) - 1.0)
    width = np.diff(pts)
    slopes = pts[None, :] - freqs[:, None]
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / width[:-1], slopes[:, 2:] / width[1:]))
    return fb.astype(np.float32)


def segment_ranges(v_len_frames: int, a_len_frames: int, v_fps: int = 25, a_fps: int = 16000, segment_size_vframes: int = 16,
                   n_segments: int = 14, step_size_seg: float = 0.5) -> dict:
    """GenerateMultipleSegments with is_start_random=False, audio_jitter_sec=0 (dataset/transforms.py:421-500; configs/sync.yaml:222-227):
    equally spaced, 50 %-overlapping segments taken from the middle of the clip.  Returns the start / stride of the video (frames)
    and audio (samples) windows instead of materialising them."""
    seg_a = int(segment_size_vframes / v_fps * a_fps)                          # sec2frames(frames2sec(.)) (:431, :12-16)
    stride_v, stride_a = int(step_size_seg * segment_size_vframes), int(step_size_seg * seg_a)
    n_max = min((v_len_frames - segment_size_vframes) // stride_v + 1, (a_len_frames - seg_a) // stride_a + 1)      # (:436-439)
    n_seg = n_max if n_segments is None else n_segments
    if n_seg > n_max:
        raise ValueError(f'cant make {n_seg} segs of len {segment_size_vframes} in a vid of len {v_len_frames}')     # (:442-444)
    seg_seq_len = n_seg * step_size_seg + (1 - step_size_seg)                 # (:467-469)
    v_start = (v_len_frames - int(seg_seq_len * segment_size_vframes)) // 2   # (:472-476)
    a_start = int(v_start / v_fps * a_fps)                                    # (:477)
    if a_start + (n_seg - 1) * stride_a + seg_a > a_len_frames:
        raise ValueError('audio ranges out of bounds')                        # (:497)
    return dict(n_segments=n_seg, v_start=v_start, v_stride=stride_v, v_size=segment_size_vframes, a_start=a_start, a_stride=stride_a,
                a_size=seg_a)


class MelFrontend:
    def __init__(self, device, sample_rate=16000, n_mels=128, pad_to=66, mean=AST_MEAN, std=AST_STD):
        self.dev = torch.device(device)
        if self.dev.type != 'cuda':
            raise RuntimeError('MelFrontend runs on a HIP device only (no CPU fallback)')
        self.n_mels, self.pad_to, self.mean, self.std, self.hop = n_mels, pad_to, float(mean), float(std), 160
        win, n_fft, bins = 400, 1024, 513
        n = np.arange(win)
        hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win)                      # periodic Hann
        ang = 2.0 * np.pi * np.outer((n_fft - win) // 2 + n, np.arange(bins)) / n_fft
        self.tw_cos = torch.from_numpy((hann[:, None] * np.cos(ang)).astype(np.float32)).to(self.dev)
        self.tw_sin = torch.from_numpy((-hann[:, None] * np.sin(ang)).astype(np.float32)).to(self.dev)
        fb = mel_filterbank(bins, 0.0, sample_rate / 2, n_mels, sample_rate)
        nz = fb > 0
        lo = np.where(nz.any(0), nz.argmax(0), 0).astype(np.int32)
        hi = np.where(nz.any(0), bins - nz[::-1].argmax(0), 0).astype(np.int32)
        self.fb = torch.from_numpy(fb).to(self.dev)
        self.fb_lo, self.fb_hi = torch.from_numpy(lo).to(self.dev), torch.from_numpy(hi).to(self.dev)
        self._ws = None

    def __call__(self, wave: torch.Tensor) -> torch.Tensor:
        """wave (..., n_samples) fp32 on device -> (..., 1, n_mels, pad_to) fp32 (PermuteStreams 'S F T -> S 1 F T')."""
        if not wave.is_cuda:
            raise RuntimeError('MelFrontend: expected a HIP device tensor (no CPU fallback exists)')
        lead, n = wave.shape[:-1], wave.shape[-1]
        w = wave.reshape(-1, n).to(torch.float32).contiguous()
        n_seg = w.shape[0]
        frames = min(n // self.hop + 1, self.pad_to)
        need = n_seg * frames * 513
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.dev, dtype=torch.float32)
        out = torch.empty(n_seg, self.n_mels, self.pad_to, device=self.dev, dtype=torch.float32)
        rc = _lib.load().sf_mel_frontend(w.data_ptr(), n_seg, n, self.hop, self.tw_cos.data_ptr(), self.tw_sin.data_ptr(),
                                         self.fb.data_ptr(), self.fb_lo.data_ptr(), self.fb_hi.data_ptr(), self.n_mels,
                                         self._ws.data_ptr(), out.data_ptr(), self.pad_to, self.mean, self.std,
                                         torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'sf_mel_frontend')
        return out.reshape(*lead, 1, self.n_mels, self.pad_to)


    def segments(self, wave: torch.Tensor, a_start: int, a_stride: int, n_seg: int, a_size: int) -> torch.Tensor:
        """wave (B, clip_samples) fp32 on device -> (B, n_seg, 1, n_mels, pad_to): the log-mel of every overlapping segment window
        [a_start + s*a_stride, +a_size), read in place from the clip (no (B, S, a_size) copy)."""
        if not wave.is_cuda:
            raise RuntimeError('MelFrontend: expected a HIP device tensor (no CPU fallback exists)')
        w = wave.to(torch.float32).contiguous()
        B, clip = w.shape
        frames = min(a_size // self.hop + 1, self.pad_to)
        need = B * n_seg * frames * 513
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, device=self.dev, dtype=torch.float32)
        out = torch.empty(B * n_seg, self.n_mels, self.pad_to, device=self.dev, dtype=torch.float32)
        rc = _lib.load().sf_mel_frontend_clips(w.data_ptr(), B, clip, a_start, a_stride, n_seg, a_size, self.hop, self.tw_cos.data_ptr(),
                                               self.tw_sin.data_ptr(), self.fb.data_ptr(), self.fb_lo.data_ptr(), self.fb_hi.data_ptr(),
                                               self.n_mels, self._ws.data_ptr(), out.data_ptr(), self.pad_to, self.mean, self.std,
                                               torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, 'sf_mel_frontend_clips')
        return out.reshape(B, n_seg, 1, self.n_mels, self.pad_to)


class HostClipPipeline:
    """Raw clips from HOST memory to logits, with the host-to-device transfer of batch i+1 overlapped with the forward of batch i.

    The reference's step starts at `prepare_inputs` (scripts/train_utils.py:359-369: `batch['video']` / `batch['audio']` -> device) behind a CPU
    DataLoader that has already segmented, normalised and mel-transformed the clip (dataset/transforms.py:402-499, 647-669, 815-889).  Here the host
    hands over what the decoder produced - uint8 frames (B, T, 3, 224, 224) and the 16 kHz waveform (B, n_samples) fp32 - in PINNED buffers; the
    transfer runs on a copy stream into one of two device slots while the compute stream works on the other (`engine.forward_clips`: segments read
    in place, RGB normalisation inside the patch gather, log-mel on the device).  The 50 %-overlapping segments are never materialised, so a 14-segment
    clip crosses PCIe as 125 frames (18.8 MB) instead of 224 (33.7 MB as uint8, 67.4 MB as the reference's fp16).

        pipe = HostClipPipeline(engine, MelFrontend(dev), B, T, n_samples)
        pipe.stage(frames0, wave0)                      # batch 0 starts moving
        for next_frames, next_wave in batches[1:]:
            logits = pipe.step(next_frames, next_wave)  # forward of the staged batch || H2D of the next one
        logits = pipe.step()                            # last batch
    The logits of a step are valid on the compute (= current) stream; a slot is overwritten only after the forward that read it has finished
    (event hand-off in both directions, no host synchronisation inside step()).

    LIFETIME OF THE HOST BUFFERS: stage() / step() only ENQUEUE the host-to-device copies (non_blocking from pinned memory), so the caller's frames_host /
    wave_host must stay untouched until that transfer has completed.  stage() returns the event recorded behind the two copies (also `pipe.loaded[slot]`);
    a loader that recycles its pinned buffers calls `pipe.wait_staged()` (host-side wait for the most recent stage()) or `event.synchronize()` /
    `event.query()` on the returned event before it overwrites them."""

    def __init__(self, engine, mel: 'MelFrontend', B: int, T: int = 125, n_samples: int = 80000, H: int = 224, W: int = 224, **segment_kw):
        self.eng, self.mel, self.dev = engine, mel, engine.dev
        self.segment_kw = segment_kw
        self.frames = [torch.empty(B, T, 3, H, W, device=self.dev, dtype=torch.uint8) for _ in range(2)]
        self.wave = [torch.empty(B, n_samples, device=self.dev, dtype=torch.float32) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=self.dev)
        self.loaded = [torch.cuda.Event(), torch.cuda.Event()]             # H2D into slot i done (recorded on the copy stream)
        self.released = [torch.cuda.Event(), torch.cuda.Event()]           # forward reading slot i done (recorded on the compute stream)
        self._fresh = [True, True]                                         # slot never read yet: nothing to wait for
        self._staged = None                                                # slot holding the batch the next step() consumes
        self._next = 0

    @staticmethod
    def pinned_like(frames: torch.Tensor, wave: torch.Tensor):
        """Pinned host copies of a batch (what a DataLoader with pin_memory=True hands over)."""
        return frames.contiguous().pin_memory(), wave.to(torch.float32).contiguous().pin_memory()

    def stage(self, frames_host: torch.Tensor, wave_host: torch.Tensor) -> 'torch.cuda.Event':
        """Start the transfer of one batch into the free slot (non-blocking for pinned sources); returns the event recorded behind the copies."""
        if self._staged is not None and self._next == self._staged:
            raise RuntimeError('HostClipPipeline: both slots are in use - call step() before staging another batch')
        if frames_host.shape != self.frames[0].shape or wave_host.shape != self.wave[0].shape or frames_host.dtype != torch.uint8 or wave_host.dtype != torch.float32:
            raise ValueError(f'HostClipPipeline: expected uint8 frames {tuple(self.frames[0].shape)} and fp32 wave {tuple(self.wave[0].shape)} '
                             f'(got {frames_host.dtype} {tuple(frames_host.shape)}, {wave_host.dtype} {tuple(wave_host.shape)}; a non-fp32 wave would take a converting copy)')
        i = self._next
        with torch.cuda.stream(self.copy_stream):
            if not self._fresh[i]:
                self.copy_stream.wait_event(self.released[i])              # the forward that last read this slot
            self.frames[i].copy_(frames_host, non_blocking=True)
            self.wave[i].copy_(wave_host, non_blocking=True)
            self.loaded[i].record(self.copy_stream)
        if self._staged is None:
            self._staged = i
        self._next = i ^ 1
        self._last_loaded = self.loaded[i]
        return self.loaded[i]

    def wait_staged(self):
        """Block the HOST until the most recent stage() has left the caller's host buffers (they may be recycled afterwards)."""
        ev = getattr(self, '_last_loaded', None)
        if ev is not None:
            ev.synchronize()

    def step(self, next_frames_host: torch.Tensor = None, next_wave_host: torch.Tensor = None) -> torch.Tensor:
        """Forward of the staged batch; if a next batch is given its transfer is issued FIRST so that it runs under this forward."""
        if self._staged is None:
            raise RuntimeError('HostClipPipeline: nothing staged')
        i = self._staged
        if next_frames_host is not None:
            self._next = i ^ 1
            self.stage(next_frames_host, next_wave_host)
        main = torch.cuda.current_stream(self.dev)
        main.wait_event(self.loaded[i])
        logits = self.eng.forward_clips(self.frames[i], self.wave[i], self.mel, **self.segment_kw)
        self.released[i].record(main)
        self._fresh[i] = False
        self._staged = (i ^ 1) if next_frames_host is not None else None
        return logits
