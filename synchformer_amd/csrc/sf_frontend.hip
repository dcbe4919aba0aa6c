// Audio front-end on device: waveform segments -> normalised log-mel spectrograms, i.e. the
// deterministic tail of the reference's CPU transform pipeline
//   AudioMelSpectrogram -> AudioLog -> PadOrTruncate -> AudioNormalizeAST -> PermuteStreams
// (dataset/transforms.py:815-889, parameters configs/sync.yaml:183-202): torchaudio MelSpectrogram(sample_rate 16000,
// win_length 400, hop 160, n_fft 1024, n_mels 128; periodic Hann zero-padded to n_fft, center=True/reflect, power 2, HTK mel
// scale, no filter normalisation) -> log(x + 1e-6) -> right-pad the time axis with 0.0 -> (x - mean) / (2 std).
// ~0.75 GFLOP per clip: latency/bandwidth class, so plain fp32 VALU with all tables (twiddles, filterbank) passed in by
// the host.  fp32 throughout - there is no bf16 anywhere in this path.
#include "sf_common.h"
#include "../../include/synchformer_hip.h"

#define MEL_WIN 400
#define MEL_NFFT 1024
#define MEL_BINS 513
#define MEL_FT 13          // frames per workgroup
#define MEL_BT 171         // bins per workgroup (3 x 171 = 513), one thread per bin

// P[seg][t][k] = | sum_m hann[m] * x[reflect(t*hop - 200 + m)] * exp(-2 pi i k (312 + m) / 1024) |^2
// tw_cos / tw_sin: [MEL_WIN][MEL_BINS] fp32 with the window already multiplied in.
__global__ __launch_bounds__(192) void mel_stft_power_kernel(const float* __restrict__ wave, int n_samples, int hop, int n_frames,
                                                              const float* __restrict__ tw_cos, const float* __restrict__ tw_sin,
                                                              float* __restrict__ P, int n_seg_clip, int64_t clip_samples, int64_t sample0,
                                                              int64_t seg_stride) {
  __shared__ float xs[MEL_FT * 160 + MEL_WIN];     // the samples this frame group touches (<= 13*160 + 400 = 2480)
  const int seg = blockIdx.z, fgp = blockIdx.y, bgp = blockIdx.x;
  const int t0 = fgp * MEL_FT;
  const int nt = min(MEL_FT, n_frames - t0);
  const int span = (nt - 1) * hop + MEL_WIN;
  // segment seg = (clip, s) starts at sample clip * clip_samples + sample0 + s * seg_stride; reflection stays inside the segment
  const float* w = wave + (int64_t)(seg / n_seg_clip) * clip_samples + sample0 + (int64_t)(seg % n_seg_clip) * seg_stride;
  for (int i = threadIdx.x; i < span; i += blockDim.x) {
    int idx = t0 * hop - (MEL_NFFT / 2 - (MEL_NFFT - MEL_WIN) / 2) + i;       // t*hop - 512 + 312 + m
    if (idx < 0) idx = -idx;                                                  // reflect padding (center=True)
    if (idx >= n_samples) idx = 2 * (n_samples - 1) - idx;
    xs[i] = w[idx];
  }
  __syncthreads();
  const int k = bgp * MEL_BT + threadIdx.x;
  if (threadIdx.x >= MEL_BT || k >= MEL_BINS) return;
  float re[MEL_FT], im[MEL_FT];
#pragma unroll
  for (int f = 0; f < MEL_FT; ++f) { re[f] = 0.f; im[f] = 0.f; }
  for (int m = 0; m < MEL_WIN; ++m) {
    const float c = tw_cos[m * MEL_BINS + k], s = tw_sin[m * MEL_BINS + k];   // coalesced across the bin threads
#pragma unroll
    for (int f = 0; f < MEL_FT; ++f) {
      const float x = xs[f * hop + m];                                         // LDS broadcast
      re[f] = fmaf(x, c, re[f]);
      im[f] = fmaf(x, s, im[f]);
    }
  }
#pragma unroll
  for (int f = 0; f < MEL_FT; ++f)
    if (f < nt) P[((int64_t)seg * n_frames + t0 + f) * MEL_BINS + k] = re[f] * re[f] + im[f] * im[f];
}

// out[seg][j][t] = (log(sum_k P[seg][t][k] * fb[k][j] + 1e-6) - mean) / (2 std) for t < n_frames; (0 - mean)/(2 std) for the
// padded frames t in [n_frames, pad_to).  fb_lo/fb_hi give each mel filter's non-zero bin range (triangular filters).
__global__ __launch_bounds__(128) void mel_log_norm_kernel(const float* __restrict__ P, int n_frames, const float* __restrict__ fb,
                                                            const int* __restrict__ fb_lo, const int* __restrict__ fb_hi, int n_mels,
                                                            float* __restrict__ out, int pad_to, float mean, float inv_two_std) {
  const int seg = blockIdx.y, t = blockIdx.x, j = threadIdx.x;
  if (j >= n_mels) return;
  float v = 0.0f;                                                              // PadOrTruncate pad value (after the log)
  if (t < n_frames) {
    const float* p = P + ((int64_t)seg * n_frames + t) * MEL_BINS;
    float acc = 0.f;
    for (int k = fb_lo[j]; k < fb_hi[j]; ++k) acc = fmaf(p[k], fb[k * n_mels + j], acc);
    v = logf(acc + 1e-6f);
  }
  out[((int64_t)seg * n_mels + j) * pad_to + t] = (v - mean) * inv_two_std;
}

static int launch_mel(const float* wave, int64_t n_seg, int n_samples, int hop, const float* tw_cos, const float* tw_sin, const float* fb,
                      const int* fb_lo, const int* fb_hi, int n_mels, float* power_ws, float* out, int pad_to, float mean, float std,
                      int n_seg_clip, int64_t clip_samples, int64_t sample0, int64_t seg_stride, hipStream_t s, const char* who) {
  SF_CHECK_ARG(wave && tw_cos && tw_sin && fb && fb_lo && fb_hi && power_ws && out, "%s: null pointer", who);
  SF_CHECK_ARG(hop == 160, "%s: hop %d unsupported (160)", who, hop);
  SF_CHECK_ARG(n_mels > 0 && n_mels <= 128, "%s: n_mels %d out of range", who, n_mels);
  SF_CHECK_ARG(n_samples >= MEL_NFFT / 2 + 1, "%s: segment shorter than the reflect padding", who);
  const int n_frames = n_samples / hop + 1;
  SF_CHECK_ARG(pad_to >= 1, "%s: bad pad_to", who);
  if (n_seg <= 0) return 0;
  SF_CHECK_ARG(n_seg < 65536, "%s: at most 65535 segments per call", who);
  const int use_frames = n_frames < pad_to ? n_frames : pad_to;               // PadOrTruncate truncates longer inputs
  dim3 g1(3, (use_frames + MEL_FT - 1) / MEL_FT, (unsigned)n_seg);
  hipLaunchKernelGGL(mel_stft_power_kernel, g1, dim3(192), 0, s, wave, n_samples, hop, use_frames, tw_cos, tw_sin, power_ws, n_seg_clip, clip_samples,
                     sample0, seg_stride);
  SF_LAUNCH_CHECK();
  dim3 g2((unsigned)pad_to, (unsigned)n_seg);
  hipLaunchKernelGGL(mel_log_norm_kernel, g2, dim3(128), 0, s, power_ws, use_frames, fb, fb_lo, fb_hi, n_mels, out, pad_to, mean,
                     1.0f / (2.0f * std));
  SF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sf_mel_frontend(const float* wave, int64_t n_seg, int n_samples, int hop, const float* tw_cos, const float* tw_sin,
                               const float* fb, const int* fb_lo, const int* fb_hi, int n_mels, float* power_ws, float* out,
                               int pad_to, float mean, float std, void* stream) {
  return launch_mel(wave, n_seg, n_samples, hop, tw_cos, tw_sin, fb, fb_lo, fb_hi, n_mels, power_ws, out, pad_to, mean, std, 1, n_samples, 0, 0,
                    (hipStream_t)stream, "sf_mel_frontend");
}

extern "C" int sf_mel_frontend_clips(const float* wave, int64_t n_clips, int64_t clip_samples, int64_t sample0, int64_t seg_stride, int n_seg,
                                     int n_samples, int hop, const float* tw_cos, const float* tw_sin, const float* fb, const int* fb_lo,
                                     const int* fb_hi, int n_mels, float* power_ws, float* out, int pad_to, float mean, float std, void* stream) {
  SF_CHECK_ARG(n_seg >= 1 && sample0 >= 0 && seg_stride >= 0 && sample0 + (int64_t)(n_seg - 1) * seg_stride + n_samples <= clip_samples,
               "sf_mel_frontend_clips: segments [%lld + s*%lld, +%d) do not fit %lld samples", (long long)sample0, (long long)seg_stride, n_samples,
               (long long)clip_samples);
  return launch_mel(wave, n_clips * n_seg, n_samples, hop, tw_cos, tw_sin, fb, fb_lo, fb_hi, n_mels, power_ws, out, pad_to, mean, std, n_seg,
                    clip_samples, sample0, seg_stride, (hipStream_t)stream, "sf_mel_frontend_clips");
}
