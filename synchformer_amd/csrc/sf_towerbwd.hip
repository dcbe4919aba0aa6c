// Backward-pass helpers for the feature-extractor towers (Stage-1 AVCLIP training):
//   * row gather / scatter of bf16 token matrices with row maps: builds the per-group sequences
//     [CLS; group tokens] of Motionformer's divided attention (vit_helper.py:100-158) so the batched-GEMM attention
//     backward can treat time groups (9 keys) and space groups (197 keys) as ordinary contiguous sequences, and
//     scatters the resulting dq|dk|dv rows back to token order;
//   * sum over groups of the CLS key/value gradient (the CLS row is a key of every group);
//   * backward of the single-query attention (Motionformer's CLS query over all 1569 tokens, vit_helper.py:126, and the
//     aggregator layers' row-0 query, motionformer.py:329-332);
//   * backward of AveragePooling 'BS t D -> BS D' + F.normalize (open_clip/model.py:530-531).
// These are all HBM/latency-bound VALU kernels (no GEMM shape in them); the matmul-shaped parts of the backward run on
// sf_gemm_bf16 / sf_gemm_bf16_batched.
#include "sf_common.h"

// ---- row gather / scatter --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_rows_bf16_kernel(const bf16_t* __restrict__ src, int64_t ld_src, RowMap sm, bf16_t* __restrict__ dst,
                                                             int64_t ld_dst, RowMap dm, int64_t rows, int chunks) {
  const int64_t r = blockIdx.x;
  const bf16_t* s = src + map_row(sm, r) * ld_src;
  bf16_t* d = dst + map_row(dm, r) * ld_dst;
  for (int c = threadIdx.x; c < chunks; c += 256) *(uint4*)(d + c * 8) = *(const uint4*)(s + c * 8);
}

extern "C" int sf_copy_rows_bf16(const uint16_t* src, int64_t ld_src, const int64_t* src_map, uint16_t* dst, int64_t ld_dst, const int64_t* dst_map,
                                 int64_t rows, int cols, void* stream) {
  SF_CHECK_ARG(src && dst && rows >= 1 && rows < (1ll << 31) && cols >= 8 && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0,
               "sf_copy_rows_bf16: bad arguments (cols, strides %% 8 == 0)");
  hipLaunchKernelGGL(copy_rows_bf16_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, src, ld_src, sf_rowmap(src_map), dst, ld_dst,
                     sf_rowmap(dst_map), rows, cols / 8);
  SF_LAUNCH_CHECK();
  return 0;
}

// out[s * out_seq_stride + c] (=|+=) sum_{g < G} in[s * in_seq_stride + g * in_group_stride + c]   (strides in elements)
__global__ __launch_bounds__(256) void reduce_groups_bf16_kernel(const bf16_t* __restrict__ in, int64_t in_seq_stride, int64_t in_group_stride, int G,
                                                                 bf16_t* __restrict__ out, int64_t out_seq_stride, int cols, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const bf16_t* p = in + (int64_t)blockIdx.y * in_seq_stride + c;
  float acc = 0.f;
  for (int g = 0; g < G; ++g) acc += bf2f(p[g * in_group_stride]);
  bf16_t* o = out + (int64_t)blockIdx.y * out_seq_stride + c;
  if (accumulate) acc += bf2f(*o);
  *o = f2bf(acc);
}

extern "C" int sf_reduce_groups_bf16(const uint16_t* in, int64_t in_seq_stride, int64_t in_group_stride, int G, uint16_t* out, int64_t out_seq_stride,
                                     int cols, int64_t n_seq, int accumulate, void* stream) {
  SF_CHECK_ARG(in && out && G >= 1 && cols >= 1 && n_seq >= 1 && n_seq < 65536, "sf_reduce_groups_bf16: bad arguments");
  hipLaunchKernelGGL(reduce_groups_bf16_kernel, dim3((cols + 255) / 256, (unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, in, in_seq_stride,
                     in_group_stride, G, out, out_seq_stride, cols, accumulate);
  SF_LAUNCH_CHECK();
  return 0;
}

// ---- single-query attention backward -----------------------------------------------------------------------------------
// One workgroup per (sequence, head), head_dim 64.  q row, dO row: one per sequence; keys/values: n_keys consecutive rows.
//   s_j = scale <q, k_j>, p = softmax(s), o = sum_j p_j v_j
//   dp_j = <dO, v_j>, ds_j = p_j (dp_j - sum_i p_i dp_i)
//   dq = scale sum_j ds_j k_j        dk_j (=|+=) scale ds_j q        dv_j (=|+=) p_j dO
#define CLSB_MAX_KEYS 2048
__global__ __launch_bounds__(256) void attention_cls_bwd_kernel(const bf16_t* __restrict__ q, int64_t q_seq_rows, int q_row, const bf16_t* __restrict__ k,
                                                                const bf16_t* __restrict__ v, int64_t ld, int64_t kv_seq_rows, int kv_row0, int n_keys,
                                                                const bf16_t* __restrict__ dO, int64_t lddo, int64_t do_seq_rows, int do_row,
                                                                bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t ldg,
                                                                int heads, float scale, int accumulate_kv) {
  __shared__ float qs[64], dos[64], s_l[CLSB_MAX_KEYS], dp_l[CLSB_MAX_KEYS], red[8], dq_part[4][64];
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t qr = (int64_t)seq * q_seq_rows + q_row;
  const int64_t kr0 = (int64_t)seq * kv_seq_rows + kv_row0;
  if (tid < 64) qs[tid] = bf2f(q[qr * ld + h * 64 + tid]);
  else if (tid < 128) dos[tid - 64] = bf2f(dO[((int64_t)seq * do_seq_rows + do_row) * lddo + h * 64 + (tid - 64)]);
  __syncthreads();
  // pass 1: scores and dp, thread per key
  float mx = -INFINITY;
  for (int j = tid; j < n_keys; j += 256) {
    const bf16_t* kp = k + (kr0 + j) * ld + h * 64;
    const bf16_t* vp = v + (kr0 + j) * ld + h * 64;
    float s = 0.f, dp = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 kk = *(const uint4*)(kp + c * 8), vv = *(const uint4*)(vp + c * 8);
      const uint32_t kw[4] = {kk.x, kk.y, kk.z, kk.w}, vw[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s += qs[c * 8 + 2 * e] * __uint_as_float(kw[e] << 16) + qs[c * 8 + 2 * e + 1] * __uint_as_float(kw[e] & 0xffff0000u);
        dp += dos[c * 8 + 2 * e] * __uint_as_float(vw[e] << 16) + dos[c * 8 + 2 * e + 1] * __uint_as_float(vw[e] & 0xffff0000u);
      }
    }
    s *= scale;
    s_l[j] = s; dp_l[j] = dp;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f, pd = 0.f;
  for (int j = tid; j < n_keys; j += 256) {
    const float p = __expf(s_l[j] - mx);
    s_l[j] = p;
    sum += p; pd += p * dp_l[j];
  }
  sum = wave_sum(sum); pd = wave_sum(pd);
  __syncthreads();                                   // everyone has read red[0..3]
  if (lane == 0) { red[wave] = sum; red[4 + wave] = pd; }
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1]) + (red[2] + red[3]));
  const float Dsum = ((red[4] + red[5]) + (red[6] + red[7])) * inv;      // sum_i p_i dp_i
  // pass 2: per-key gradients; dq partials in registers
  float dqa[64];
#pragma unroll
  for (int d = 0; d < 64; ++d) dqa[d] = 0.f;
  for (int j = tid; j < n_keys; j += 256) {
    const float p = s_l[j] * inv;
    const float ds = p * (dp_l[j] - Dsum) * scale;
    const bf16_t* kp = k + (kr0 + j) * ld + h * 64;
    bf16_t* dkp = dk + (kr0 + j) * ldg + h * 64;
    bf16_t* dvp = dv + (kr0 + j) * ldg + h * 64;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 kk = *(const uint4*)(kp + c * 8);
      const uint32_t kw[4] = {kk.x, kk.y, kk.z, kk.w};
      float gk[8], gv[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        dqa[c * 8 + 2 * e] += ds * __uint_as_float(kw[e] << 16);
        dqa[c * 8 + 2 * e + 1] += ds * __uint_as_float(kw[e] & 0xffff0000u);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { gk[e] = ds * qs[c * 8 + e]; gv[e] = p * dos[c * 8 + e]; }
      if (accumulate_kv) {
        const uint4 ok = *(const uint4*)(dkp + c * 8), ov = *(const uint4*)(dvp + c * 8);
        const uint32_t okw[4] = {ok.x, ok.y, ok.z, ok.w}, ovw[4] = {ov.x, ov.y, ov.z, ov.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          gk[2 * e] += __uint_as_float(okw[e] << 16); gk[2 * e + 1] += __uint_as_float(okw[e] & 0xffff0000u);
          gv[2 * e] += __uint_as_float(ovw[e] << 16); gv[2 * e + 1] += __uint_as_float(ovw[e] & 0xffff0000u);
        }
      }
      uint4 wk, wv;
      wk.x = pack_bf2(gk[0], gk[1]); wk.y = pack_bf2(gk[2], gk[3]); wk.z = pack_bf2(gk[4], gk[5]); wk.w = pack_bf2(gk[6], gk[7]);
      wv.x = pack_bf2(gv[0], gv[1]); wv.y = pack_bf2(gv[2], gv[3]); wv.z = pack_bf2(gv[4], gv[5]); wv.w = pack_bf2(gv[6], gv[7]);
      *(uint4*)(dkp + c * 8) = wk;
      *(uint4*)(dvp + c * 8) = wv;
    }
  }
#pragma unroll
  for (int d = 0; d < 64; ++d) {
    const float t = wave_sum(dqa[d]);
    if (lane == 0) dq_part[wave][d] = t;
  }
  __syncthreads();
  if (tid < 64) dq[qr * ldg + h * 64 + tid] = f2bf((dq_part[0][tid] + dq_part[1][tid]) + (dq_part[2][tid] + dq_part[3][tid]));
}

extern "C" int sf_attention_cls_bwd(const uint16_t* q, int64_t q_seq_rows, int q_row, const uint16_t* k, const uint16_t* v, int64_t ld,
                                    int64_t kv_seq_rows, int kv_row0, int n_keys, const uint16_t* dO, int64_t lddo, int64_t do_seq_rows, int do_row,
                                    uint16_t* dq, uint16_t* dk, uint16_t* dv, int64_t ldg, int64_t n_seq, int heads, int head_dim, float scale,
                                    int accumulate_kv, void* stream) {
  SF_CHECK_ARG(q && k && v && dO && dq && dk && dv && head_dim == 64 && n_keys >= 1 && n_keys <= CLSB_MAX_KEYS && heads >= 1 && n_seq >= 1 &&
                   ld % 8 == 0 && ldg % 8 == 0 && n_seq * heads < (1ll << 31),
               "sf_attention_cls_bwd: bad arguments (head_dim 64, n_keys <= %d, strides %% 8 == 0)", CLSB_MAX_KEYS);
  hipLaunchKernelGGL(attention_cls_bwd_kernel, dim3((unsigned)(n_seq * heads)), dim3(256), 0, (hipStream_t)stream, q, q_seq_rows, q_row, k, v, ld,
                     kv_seq_rows, kv_row0, n_keys, dO, lddo, do_seq_rows, do_row, dq, dk, dv, ldg, heads, scale, accumulate_kv);
  SF_LAUNCH_CHECK();
  return 0;
}

// ---- AveragePooling + F.normalize backward ---------------------------------------------------------------------------------
// forward: m = mean_j x[r*t + j], y = normalize ? m / max(||m||, 1e-12) : m.   dx[r*t + j] = dm / t with
//   dm = normalize ? (dy - y <y, dy>) / ||m|| : dy.   One wave per pooled row.
__global__ __launch_bounds__(256) void meanpool_l2norm768_bwd_kernel(const float* __restrict__ x, int64_t ldx, int t, const float* __restrict__ dy,
                                                                     int64_t lddy, float* __restrict__ dx, int64_t lddx, int normalize, int64_t n) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n) return;
  float4 g[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) g[i] = *(const float4*)(dy + row * lddy + lane * 4 + 256 * i);
  const float inv_t = 1.0f / (float)t;
  if (normalize) {
    float4 m[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) m[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < t; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float4 vv = *(const float4*)(x + (row * t + j) * ldx + lane * 4 + 256 * i);
        m[i].x += vv.x; m[i].y += vv.y; m[i].z += vv.z; m[i].w += vv.w;
      }
    float ss = 0.f, dot = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      m[i].x *= inv_t; m[i].y *= inv_t; m[i].z *= inv_t; m[i].w *= inv_t;
      ss += m[i].x * m[i].x + m[i].y * m[i].y + m[i].z * m[i].z + m[i].w * m[i].w;
      dot += m[i].x * g[i].x + m[i].y * g[i].y + m[i].z * g[i].z + m[i].w * g[i].w;
    }
    ss = wave_sum(ss); dot = wave_sum(dot);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    const float a = 1.0f / nrm, b = dot / (nrm * nrm * nrm);          // dm = dy / ||m|| - m <m, dy> / ||m||^3
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      g[i].x = g[i].x * a - m[i].x * b; g[i].y = g[i].y * a - m[i].y * b;
      g[i].z = g[i].z * a - m[i].z * b; g[i].w = g[i].w * a - m[i].w * b;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) { g[i].x *= inv_t; g[i].y *= inv_t; g[i].z *= inv_t; g[i].w *= inv_t; }
  for (int j = 0; j < t; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) *(float4*)(dx + (row * t + j) * lddx + lane * 4 + 256 * i) = g[i];
}

extern "C" int sf_meanpool_l2norm768_bwd(const float* x, int64_t ldx, int t, const float* dy, int64_t lddy, float* dx, int64_t lddx, int normalize,
                                         int64_t n, void* stream) {
  SF_CHECK_ARG(x && dy && dx && t >= 1 && n >= 1 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0 && ldx >= 768 && lddy >= 768 && lddx >= 768,
               "sf_meanpool_l2norm768_bwd: bad arguments");
  hipLaunchKernelGGL(meanpool_l2norm768_bwd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, t, dy, lddy, dx, lddx,
                     normalize, n);
  SF_LAUNCH_CHECK();
  return 0;
}
