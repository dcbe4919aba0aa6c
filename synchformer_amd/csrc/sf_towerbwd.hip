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

// ---- the SIDE rows of the fused attention launches in one gather ----------------------------------------------------------------------------------------------
// sf_qkv_space_attention / sf_qkv_time_attention2 do not project 33 rows of every sequence themselves - the CLS row and tokens 192..195 of each of the 8 frames (196 = 6 x 32 + 4
// = 8 x 24 + 4); they go through one small GEMM on a gathered copy.  One launch, a workgroup per destination row: row seq * 33 <- X row seq * seq_rows, row seq * 33 + 1 + 4 f
// + i <- X row seq * seq_rows + 1 + n_tok f + (n_tok - 4) + i; row_bytes per row (1536: bf16; 768: the e4m3 bytes of an MXFP8 operand), and with sX the rows' E8M0 scale
// dwords of every stage-major plane as well (plane k: sX + k * ldsx -> sO + k * ldso, one dword per row).  Was two sf_copy_rows_bf16 launches (bf16) / three torch index ops
// per half block (MXFP8).
__global__ __launch_bounds__(128) void side_rows_kernel(const char* __restrict__ X, int64_t ldx, char* __restrict__ out, int64_t ldo, int chunks, const char* __restrict__ sX,
                                                        int64_t ldsx, char* __restrict__ sO, int64_t ldso, int n_planes, int64_t seq_rows, int n_tok) {
  const int64_t d = blockIdx.x, seq = d / 33;
  const int r = (int)(d - seq * 33);
  const int64_t srow = seq * seq_rows + (r == 0 ? 0 : 1 + (int64_t)((r - 1) >> 2) * n_tok + (n_tok - 4) + ((r - 1) & 3));
  const char* s = X + srow * ldx;
  char* o = out + d * ldo;
  for (int c = threadIdx.x; c < chunks; c += 128) *(uint4*)(o + c * 16) = *(const uint4*)(s + c * 16);
  if (sX && (int)threadIdx.x < n_planes) *(uint32_t*)(sO + (int64_t)threadIdx.x * ldso + d * 4) = *(const uint32_t*)(sX + (int64_t)threadIdx.x * ldsx + srow * 4);
}

extern "C" int sf_side_rows(const void* X, int64_t ldx_bytes, void* out, int64_t ldo_bytes, int row_bytes, const uint8_t* sX, int64_t ldsx, uint8_t* sOut, int64_t ldso,
                            int n_planes, int64_t n_seq, int n_tok, void* stream) {
  SF_CHECK_ARG(X && out && row_bytes >= 16 && (row_bytes % 16) == 0 && (ldx_bytes % 16) == 0 && (ldo_bytes % 16) == 0 && ldx_bytes >= row_bytes && ldo_bytes >= row_bytes &&
                   ((uintptr_t)X % 16) == 0 && ((uintptr_t)out % 16) == 0, "sf_side_rows: rows of whole 16-byte chunks, 16-byte aligned");
  SF_CHECK_ARG(n_tok >= 4 && n_planes >= 0 && n_planes <= 128, "sf_side_rows: bad n_tok / n_planes");
  SF_CHECK_ARG((sX == nullptr) == (sOut == nullptr) && (!sX || (n_planes >= 1 && (ldsx % 4) == 0 && (ldso % 4) == 0 && ((uintptr_t)sX % 4) == 0 && ((uintptr_t)sOut % 4) == 0)),
               "sf_side_rows: scale planes come in pairs (source, destination), one dword per row");
  if (n_seq <= 0) return 0;
  SF_CHECK_ARG(n_seq * 33 < ((int64_t)1 << 31), "sf_side_rows: too many rows");
  hipLaunchKernelGGL(side_rows_kernel, dim3((unsigned)(n_seq * 33)), dim3(128), 0, (hipStream_t)stream, (const char*)X, ldx_bytes, (char*)out, ldo_bytes, row_bytes / 16,
                     (const char*)sX, ldsx, (char*)sOut, ldso, n_planes, 1 + 8 * (int64_t)n_tok, n_tok);
  SF_LAUNCH_CHECK();
  return 0;
}

// out[s * out_seq_stride + c] (=|+=) sum_{g < G} in[s * in_seq_stride + g * in_group_stride + c]   (strides in elements)
// One workgroup per (sequence, 512 columns): 64 column chunks of 16 bytes x 16 group slices, then a tree over the slices in LDS.  (The first version ran one thread per
// column over all G groups with 2-byte loads: 70 us for the 196 time groups' 17 MB - the launch is latency, not bytes.)  General strides / ragged columns take the
// scalar kernel.
__global__ __launch_bounds__(1024) void reduce_groups_bf16_vec_kernel(const bf16_t* __restrict__ in, int64_t in_seq_stride, int64_t in_group_stride, int G,
                                                                      bf16_t* __restrict__ out, int64_t out_seq_stride, int cols, int accumulate) {
  __shared__ float part[16][64][9];                               // [slice][chunk][8 + pad]
  const int ch = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + ch) * 8;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  if (c < cols) {
    const bf16_t* p = in + (int64_t)blockIdx.y * in_seq_stride + c;
    for (int g = sl; g < G; g += 16) {
      const uint4 u = *reinterpret_cast<const uint4*>(p + g * in_group_stride);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[2 * i] += __uint_as_float(w[i] << 16); acc[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[sl][ch][e] = acc[e];
  __syncthreads();
  if (sl == 0 && c < cols) {
#pragma unroll
    for (int s2 = 1; s2 < 16; ++s2)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += part[s2][ch][e];
    bf16_t* o = out + (int64_t)blockIdx.y * out_seq_stride + c;
    if (accumulate) {
      const uint4 u = *reinterpret_cast<const uint4*>(o);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[2 * i] += __uint_as_float(w[i] << 16); acc[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
    }
    uint4 r;
    r.x = pack_bf2(acc[0], acc[1]); r.y = pack_bf2(acc[2], acc[3]); r.z = pack_bf2(acc[4], acc[5]); r.w = pack_bf2(acc[6], acc[7]);
    *reinterpret_cast<uint4*>(o) = r;
  }
}
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
  const bool vec = (cols % 8) == 0 && (in_seq_stride % 8) == 0 && (in_group_stride % 8) == 0 && (out_seq_stride % 8) == 0 && ((uintptr_t)in % 16) == 0 &&
                   ((uintptr_t)out % 16) == 0 && G >= 8;
  if (vec) hipLaunchKernelGGL(reduce_groups_bf16_vec_kernel, dim3((cols / 8 + 63) / 64, (unsigned)n_seq), dim3(1024), 0, (hipStream_t)stream, in, in_seq_stride,
                              in_group_stride, G, out, out_seq_stride, cols, accumulate);
  else hipLaunchKernelGGL(reduce_groups_bf16_kernel, dim3((cols + 255) / 256, (unsigned)n_seq), dim3(256), 0, (hipStream_t)stream, in, in_seq_stride,
                          in_group_stride, G, out, out_seq_stride, cols, accumulate);
  SF_LAUNCH_CHECK();
  return 0;
}

// ---- single-query attention backward -----------------------------------------------------------------------------------
// One workgroup per (sequence, head), head_dim 64.  q row, dO row: one per sequence; keys/values: n_keys consecutive rows.
//   s_j = scale <q, k_j>, p = softmax(s), o = sum_j p_j v_j
//   dp_j = <dO, v_j>, ds_j = p_j (dp_j - sum_i p_i dp_i)
//   dq = scale sum_j ds_j k_j        dk_j (=|+=) scale ds_j q        dv_j (=|+=) p_j dO
// Eight lanes per key: lane (key slot = tid >> 3, chunk = tid & 7) owns 8 of the 64 head dims, so every K / V / dK / dV row is touched as
// one whole 128-byte line by 8 neighbouring lanes (the first version ran one THREAD per key: every 16-byte load of a wave hit 64
// different cache lines, eight times over, and each thread carried a 64-register dq accumulator).  Scores use v_dot2c_f32_bf16.
#define CLSB_MAX_KEYS 2048
#define CLSB_THREADS 512
typedef __attribute__((ext_vector_type(2))) __bf16 clsb_bf2;
__device__ __forceinline__ float clsb_dot8(const uint4& a, const uint4& b) {
  float d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(clsb_bf2, a.x), __builtin_bit_cast(clsb_bf2, b.x), 0.f, false);
  d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(clsb_bf2, a.y), __builtin_bit_cast(clsb_bf2, b.y), d, false);
  d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(clsb_bf2, a.z), __builtin_bit_cast(clsb_bf2, b.z), d, false);
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(clsb_bf2, a.w), __builtin_bit_cast(clsb_bf2, b.w), d, false);
}
__device__ __forceinline__ void clsb_unpack8(const uint4& u, float* f) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}

__global__ __launch_bounds__(CLSB_THREADS) void attention_cls_bwd_kernel(const bf16_t* __restrict__ q, int64_t q_seq_rows, int q_row, const bf16_t* __restrict__ k,
                                                                         const bf16_t* __restrict__ v, int64_t ld, int64_t kv_seq_rows, int kv_row0, int n_keys,
                                                                         const bf16_t* __restrict__ dO, int64_t lddo, int64_t do_seq_rows, int do_row,
                                                                         bf16_t* __restrict__ dq, bf16_t* __restrict__ dk, bf16_t* __restrict__ dv, int64_t ldg,
                                                                         int heads, float scale, int accumulate_kv) {
  constexpr int NW = CLSB_THREADS / 64, KSLOTS = CLSB_THREADS / 8;
  __shared__ float s_l[CLSB_MAX_KEYS], dp_l[CLSB_MAX_KEYS], red[2 * NW], dq_part[NW][64];
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ks = tid >> 3, ch = tid & 7;
  const int64_t qr = (int64_t)seq * q_seq_rows + q_row;
  const int64_t kr0 = (int64_t)seq * kv_seq_rows + kv_row0;
  const uint4 qraw = *(const uint4*)(q + qr * ld + h * 64 + ch * 8);                                        // this lane's 8 dims of q and dO
  const uint4 doraw = *(const uint4*)(dO + ((int64_t)seq * do_seq_rows + do_row) * lddo + h * 64 + ch * 8);
  // pass 1: scores and dp
  float mx = -INFINITY;
  for (int j = ks; j < n_keys; j += KSLOTS) {
    const uint4 kk = *(const uint4*)(k + (kr0 + j) * ld + h * 64 + ch * 8), vv = *(const uint4*)(v + (kr0 + j) * ld + h * 64 + ch * 8);
    float s = clsb_dot8(qraw, kk), dp = clsb_dot8(doraw, vv);
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    dp += __shfl_xor(dp, 1, 64); dp += __shfl_xor(dp, 2, 64); dp += __shfl_xor(dp, 4, 64);
    s *= scale;
    if (ch == 0) { s_l[j] = s; dp_l[j] = dp; }
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[w]);
  float sum = 0.f, pd = 0.f;
  for (int j = tid; j < n_keys; j += CLSB_THREADS) {
    const float p = __expf(s_l[j] - mx);
    s_l[j] = p;
    sum += p; pd += p * dp_l[j];
  }
  sum = wave_sum(sum); pd = wave_sum(pd);
  __syncthreads();                                   // everyone has read red[0..NW)
  if (lane == 0) { red[wave] = sum; red[NW + wave] = pd; }
  __syncthreads();
  float tot = 0.f, totd = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { tot += red[w]; totd += red[NW + w]; }
  const float inv = 1.0f / tot;
  const float Dsum = totd * inv;                                         // sum_i p_i dp_i
  // pass 2: per-key gradients; this lane's 8 dims of dq accumulate over its keys
  float qf[8], dof[8], dqa[8];
  clsb_unpack8(qraw, qf); clsb_unpack8(doraw, dof);
#pragma unroll
  for (int e = 0; e < 8; ++e) dqa[e] = 0.f;
  for (int j = ks; j < n_keys; j += KSLOTS) {
    const float p = s_l[j] * inv;
    const float ds = p * (dp_l[j] - Dsum) * scale;
    const uint4 kk = *(const uint4*)(k + (kr0 + j) * ld + h * 64 + ch * 8);
    bf16_t* dkp = dk + (kr0 + j) * ldg + h * 64 + ch * 8;
    bf16_t* dvp = dv + (kr0 + j) * ldg + h * 64 + ch * 8;
    float kf[8], gk[8], gv[8];
    clsb_unpack8(kk, kf);
#pragma unroll
    for (int e = 0; e < 8; ++e) { dqa[e] += ds * kf[e]; gk[e] = ds * qf[e]; gv[e] = p * dof[e]; }
    if (accumulate_kv) {
      float ok[8], ov[8];
      clsb_unpack8(*(const uint4*)dkp, ok); clsb_unpack8(*(const uint4*)dvp, ov);
#pragma unroll
      for (int e = 0; e < 8; ++e) { gk[e] += ok[e]; gv[e] += ov[e]; }
    }
    uint4 wk, wv;
    wk.x = pack_bf2(gk[0], gk[1]); wk.y = pack_bf2(gk[2], gk[3]); wk.z = pack_bf2(gk[4], gk[5]); wk.w = pack_bf2(gk[6], gk[7]);
    wv.x = pack_bf2(gv[0], gv[1]); wv.y = pack_bf2(gv[2], gv[3]); wv.z = pack_bf2(gv[4], gv[5]); wv.w = pack_bf2(gv[6], gv[7]);
    *(uint4*)dkp = wk;
    *(uint4*)dvp = wv;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {                                           // over the 8 key slots of the wave that share this chunk
    float t = dqa[e];
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    if (lane < 8) dq_part[wave][ch * 8 + e] = t;
  }
  __syncthreads();
  if (tid < 64) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += dq_part[w][tid];
    dq[qr * ldg + h * 64 + tid] = f2bf(t);
  }
}

extern "C" int sf_attention_cls_bwd(const uint16_t* q, int64_t q_seq_rows, int q_row, const uint16_t* k, const uint16_t* v, int64_t ld,
                                    int64_t kv_seq_rows, int kv_row0, int n_keys, const uint16_t* dO, int64_t lddo, int64_t do_seq_rows, int do_row,
                                    uint16_t* dq, uint16_t* dk, uint16_t* dv, int64_t ldg, int64_t n_seq, int heads, int head_dim, float scale,
                                    int accumulate_kv, void* stream) {
  SF_CHECK_ARG(q && k && v && dO && dq && dk && dv && head_dim == 64 && n_keys >= 1 && n_keys <= CLSB_MAX_KEYS && heads >= 1 && n_seq >= 1 &&
                   ld % 8 == 0 && ldg % 8 == 0 && lddo % 8 == 0 && n_seq * heads < (1ll << 31),
               "sf_attention_cls_bwd: bad arguments (head_dim 64, n_keys <= %d, strides %% 8 == 0)", CLSB_MAX_KEYS);
  hipLaunchKernelGGL(attention_cls_bwd_kernel, dim3((unsigned)(n_seq * heads)), dim3(CLSB_THREADS), 0, (hipStream_t)stream, q, q_seq_rows, q_row, k, v, ld,
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
