// Stage-1 (segment-level audio-visual contrastive, AVCLIP) head: the three small fp32 ops that sit between the feature
// extractors and the symmetric cross-entropy  (train_clip_src/open_clip/model.py:449-585).
//   * mean over the t aggregated tokens of a segment (AveragePooling 'BS t D -> BS D', motionformer.py:395-409) and
//     F.normalize(dim=-1)                                                        (open_clip/model.py:530-531)
//   * the (n, m) similarity matrix  vfeat @ afeat_all^T / logit_scale              (open_clip/model.py:508-509)
// Sizes are tiny (n = B*S <= a few hundred rows of 768): these are latency/HBM-bound fp32 kernels and stay off MFMA so
// the similarity logits keep fp32 precision (they are divided by a temperature as small as 0.001).
#include "sf_common.h"

// One wave per output row; lane owns float4 columns lane*4 + 256*i.
__global__ __launch_bounds__(256) void meanpool_l2norm768_kernel(const float* __restrict__ x, int64_t ldx, int t, float* __restrict__ y, int64_t ldy,
                                                                 int normalize, int64_t n) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  float4 acc[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int j = 0; j < t; ++j) {
    const float* xr = x + (row * t + j) * ldx;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 v = *(const float4*)(xr + lane * 4 + 256 * i);
      acc[i].x += v.x; acc[i].y += v.y; acc[i].z += v.z; acc[i].w += v.w;
    }
  }
  const float inv_t = 1.0f / (float)t;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    acc[i].x *= inv_t; acc[i].y *= inv_t; acc[i].z *= inv_t; acc[i].w *= inv_t;
    ss += acc[i].x * acc[i].x + acc[i].y * acc[i].y + acc[i].z * acc[i].z + acc[i].w * acc[i].w;
  }
  float s = 1.0f;
  if (normalize) s = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);     // F.normalize: x / max(||x||_2, eps=1e-12)
#pragma unroll
  for (int i = 0; i < 3; ++i)
    *(float4*)(y + row * ldy + lane * 4 + 256 * i) = make_float4(acc[i].x * s, acc[i].y * s, acc[i].z * s, acc[i].w * s);
}

extern "C" int sf_meanpool_l2norm768(const float* x, int64_t ldx, int t, float* y, int64_t ldy, int normalize, int64_t n, void* stream) {
  SF_CHECK_ARG(x && y && t >= 1 && n >= 1 && ldx >= 768 && ldy >= 768 && ldx % 4 == 0 && ldy % 4 == 0, "sf_meanpool_l2norm768: bad arguments");
  hipLaunchKernelGGL(meanpool_l2norm768_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, t, y, ldy, normalize, n);
  SF_LAUNCH_CHECK();
  return 0;
}

// out[i, j] = scale * <a_i, b_j>: 64x64 output tile per workgroup, 4x4 per thread, 16-deep k slices through LDS
// (k-major so the inner product reads two float4s per k).
#define SIM_T 64
#define SIM_K 16
__global__ __launch_bounds__(256) void similarity_f32_kernel(const float* __restrict__ a, int64_t lda, const float* __restrict__ b, int64_t ldb,
                                                             float* __restrict__ out, int64_t ldo, int n, int m, int d, float scale) {
  __shared__ float As[SIM_K][SIM_T + 4], Bs[SIM_K][SIM_T + 4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * SIM_T, j0 = blockIdx.x * SIM_T;
  const int lr = threadIdx.x >> 2, lk = (threadIdx.x & 3) * 4;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < d; k0 += SIM_K) {
    float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
    if (i0 + lr < n) va = *(const float4*)(a + (int64_t)(i0 + lr) * lda + k0 + lk);
    if (j0 + lr < m) vb = *(const float4*)(b + (int64_t)(j0 + lr) * ldb + k0 + lk);
    As[lk + 0][lr] = va.x; As[lk + 1][lr] = va.y; As[lk + 2][lr] = va.z; As[lk + 3][lr] = va.w;
    Bs[lk + 0][lr] = vb.x; Bs[lk + 1][lr] = vb.y; Bs[lk + 2][lr] = vb.z; Bs[lk + 3][lr] = vb.w;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SIM_K; ++k) {
      const float4 av = *(const float4*)&As[k][ty * 4];
      const float4 bv = *(const float4*)&Bs[k][tx * 4];
      const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += ar[i] * br[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = i0 + ty * 4 + i;
    if (r >= n) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = j0 + tx * 4 + j;
      if (c < m) out[(int64_t)r * ldo + c] = acc[i][j] * scale;
    }
  }
}

extern "C" int sf_similarity_f32(const float* a, int64_t lda, const float* b, int64_t ldb, float* out, int64_t ldo, int n, int m, int d, float scale,
                                 void* stream) {
  SF_CHECK_ARG(a && b && out && n >= 1 && m >= 1 && d >= SIM_K && d % SIM_K == 0 && lda % 4 == 0 && ldb % 4 == 0 && lda >= d && ldb >= d && ldo >= m,
               "sf_similarity_f32: bad arguments (d %% 16 == 0, row strides %% 4 == 0)");
  hipLaunchKernelGGL(similarity_f32_kernel, dim3((m + SIM_T - 1) / SIM_T, (n + SIM_T - 1) / SIM_T), dim3(256), 0, (hipStream_t)stream, a, lda, b, ldb,
                     out, ldo, n, m, d, scale);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Stage-1 zero-shot synchronisation check (shift_and_get_preds, train_clip_src/training/train.py:549-579): for every clip the
// similarity of window i of A with window j of V, sim[i, j] = sum_{w < W} <a[i+w], v[j+w]>, is a W-long diagonal sum of the
// clip's S x S block of the audio-to-video segment similarity matrix G (rows = audio segments, cols = video segments).
//   preds_a[b, j] = argmax_i sim[i, j]   (torch.argmax(sim, dim=-2)),   preds_v[b, i] = argmax_j sim[i, j]   (dim=-1)
// One workgroup per clip; n = S - W + 1 <= 32 shifts.  First maximum wins, like torch.argmax.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shift_window_preds_kernel(const float* __restrict__ G, int64_t ldg, int S, int W, int64_t* __restrict__ preds_a,
                                                                  int64_t* __restrict__ preds_v) {
  __shared__ float sim[32][33];
  const int b = blockIdx.x, n = S - W + 1;
  const float* g = G + ((int64_t)b * S) * ldg + (int64_t)b * S;           // this clip's diagonal block
  for (int e = threadIdx.x; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    float acc = 0.f;
    for (int w = 0; w < W; ++w) acc += g[(int64_t)(i + w) * ldg + (j + w)];
    sim[i][j] = acc;
  }
  __syncthreads();
  if (threadIdx.x < n) {                                                  // column argmax -> preds_a
    const int j = threadIdx.x;
    int best = 0;
    for (int i = 1; i < n; ++i) if (sim[i][j] > sim[best][j]) best = i;
    preds_a[(int64_t)b * n + j] = best;
  } else if (threadIdx.x >= 64 && threadIdx.x < 64 + n) {                 // row argmax -> preds_v
    const int i = threadIdx.x - 64;
    int best = 0;
    for (int j = 1; j < n; ++j) if (sim[i][j] > sim[i][best]) best = j;
    preds_v[(int64_t)b * n + i] = best;
  }
}

extern "C" int sf_shift_window_preds(const float* G, int64_t ldg, int n_clips, int S, int W, int64_t* preds_a, int64_t* preds_v, void* stream) {
  SF_CHECK_ARG(G && preds_a && preds_v && n_clips >= 1 && W >= 1 && W <= S && S - W + 1 <= 32 && ldg >= (int64_t)n_clips * S,
               "sf_shift_window_preds: bad arguments (1 <= W <= S, S - W + 1 <= 32)");
  hipLaunchKernelGGL(shift_window_preds_kernel, dim3((unsigned)n_clips), dim3(256), 0, (hipStream_t)stream, G, ldg, S, W, preds_a, preds_v);
  SF_LAUNCH_CHECK();
  return 0;
}
