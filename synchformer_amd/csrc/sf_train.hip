// Kernels of the Stage-2 train step: the backward of vproj/aproj + the 3-block sync
// transformer (the only trainable part when the extractors are frozen, scripts/train_utils.py:199-204) and the fused
// clip + Adam update (train_utils.py:373-386, :217-226).  All GEMM-shaped backward work (dgrad, wgrad, the five products of
// the attention backward) goes through sf_gemm_bf16 / sf_gemm_bf16_batched on (zero-padded) transposed copies; this file
// holds the bandwidth/latency-class pieces around them.  Activations here are (B*198, 768)-sized - a few MB - so the
// kernels favour simplicity and determinism (no atomics: column reductions are two-stage).
#include "sf_common.h"
#include "../../include/synchformer_hip.h"

// ------------------------------------------------------------------------------------------------------
// Batched bf16 transpose with zero padding: out[b][c][r] = in[b][r][c] for r < R, 0 for R <= r < R_pad.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, int64_t ld_in, int64_t sI0, int64_t sI1,
                                                              bf16_t* __restrict__ out, int64_t ld_out, int64_t sO0, int64_t sO1,
                                                              int R, int C, int R_pad, int batch_inner) {
  __shared__ bf16_t tile[32][33];
  const int b0 = blockIdx.z / batch_inner, b1 = blockIdx.z - b0 * batch_inner;
  in += b0 * sI0 + b1 * sI1;
  out += b0 * sO0 + b1 * sO1;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;            // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    tile[ty + i * 8][tx] = (r < R && c < C) ? in[(int64_t)r * ld_in + c] : (bf16_t)0;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + i * 8, r = r0 + tx;
    if (c < C && r < R_pad) out[(int64_t)c * ld_out + r] = tile[tx][ty + i * 8];
  }
}

// 64 x 64 tiles, 16-byte global accesses on both sides (the 32 x 32 / 2-byte kernel above moves 64 B per wavefront row and was 20 % of
// the Stage-1 train step).  Needs C % 8 == 0, 16-byte aligned rows on both sides and R_pad % 8 == 0; everything else takes the kernel above.
__global__ __launch_bounds__(256) void transpose_bf16_kernel64(const bf16_t* __restrict__ in, int64_t ld_in, int64_t sI0, int64_t sI1,
                                                                bf16_t* __restrict__ out, int64_t ld_out, int64_t sO0, int64_t sO1,
                                                                int R, int C, int R_pad, int batch_inner) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];      // 144-B rows: 16-B aligned stores, 36-dword stride for the column gathers
  const int b0 = blockIdx.z / batch_inner, b1 = blockIdx.z - b0 * batch_inner;
  in += b0 * sI0 + b1 * sI1;
  out += b0 * sO0 + b1 * sO1;
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * 256, row = idx >> 3, ch = idx & 7;
    const int r = r0 + row, c = c0 + ch * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < R && c < C) v = *reinterpret_cast<const uint4*>(in + (int64_t)r * ld_in + c);
    *reinterpret_cast<uint4*>(&tile[row][ch * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * 256, oc = idx & 63, rch = idx >> 6;     // lanes of a wave walk the output rows: conflict-light LDS gathers
    const int c = c0 + oc, r = r0 + rch * 8;
    if (c < C && r < R_pad) {
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[rch * 8 + 2 * e][oc] | ((uint32_t)tile[rch * 8 + 2 * e + 1][oc] << 16);
      *reinterpret_cast<uint4*>(out + (int64_t)c * ld_out + r) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

extern "C" int sf_transpose_bf16(const bf16_t* in, int64_t ld_in, int64_t sI0, int64_t sI1, bf16_t* out, int64_t ld_out, int64_t sO0,
                                 int64_t sO1, int R, int C, int R_pad, int batch_outer, int batch_inner, void* stream) {
  SF_CHECK_ARG(in && out, "sf_transpose_bf16: null pointer");
  SF_CHECK_ARG(R > 0 && C > 0 && R_pad >= R && batch_outer >= 1 && batch_inner >= 1, "sf_transpose_bf16: bad shape");
  SF_CHECK_ARG((int64_t)batch_outer * batch_inner < 65536, "sf_transpose_bf16: too many batches");
  const bool wide = (C % 8) == 0 && (R_pad % 8) == 0 && (ld_in % 8) == 0 && (ld_out % 8) == 0 && (sI0 % 8) == 0 && (sI1 % 8) == 0 &&
                    (sO0 % 8) == 0 && (sO1 % 8) == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0;
  if (wide) {
    dim3 grid((R_pad + 63) / 64, (C + 63) / 64, batch_outer * batch_inner);
    hipLaunchKernelGGL(transpose_bf16_kernel64, grid, dim3(256), 0, (hipStream_t)stream, in, ld_in, sI0, sI1, out, ld_out, sO0, sO1, R, C,
                       R_pad, batch_inner);
  } else {
    dim3 grid((R_pad + 31) / 32, (C + 31) / 32, batch_outer * batch_inner);
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, ld_in, sI0, sI1, out, ld_out, sO0, sO1, R, C,
                       R_pad, batch_inner);
  }
  SF_LAUNCH_CHECK();
  return 0;
}

// Many transposes in ONE launch (the bf16 W^T copies of every trainable weight after an optimizer step: 194 launches of ~5 us in the Stage-1 step).
// table[t] = { in, out, ld_in, ld_out, R, C, R_pad, tiles_x } as int64 (device memory, built once: the operand copies never move); tile_prefix[t] = number of
// 64 x 64 tiles of tensors 0 .. t-1 (n + 1 entries).  Every tensor must satisfy the conditions of the wide kernel above.
__global__ __launch_bounds__(256) void transpose_bf16_multi_kernel(const int64_t* __restrict__ table, const int* __restrict__ tile_prefix, int n_tensors) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][72];
  int lo = 0, hi = n_tensors;                                        // the tensor whose tile range holds blockIdx.x
  const int b = blockIdx.x;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tile_prefix[mid] <= b) lo = mid; else hi = mid; }
  const int64_t* d = table + (int64_t)lo * 8;
  const bf16_t* in = reinterpret_cast<const bf16_t*>(d[0]);
  bf16_t* out = reinterpret_cast<bf16_t*>(d[1]);
  const int64_t ld_in = d[2], ld_out = d[3];
  const int R = (int)d[4], C = (int)d[5], R_pad = (int)d[6], tiles_x = (int)d[7];
  const int tl = b - tile_prefix[lo];
  const int r0 = (tl % tiles_x) * 64, c0 = (tl / tiles_x) * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * 256, row = idx >> 3, ch = idx & 7;
    const int r = r0 + row, c = c0 + ch * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < R && c < C) v = *reinterpret_cast<const uint4*>(in + (int64_t)r * ld_in + c);
    *reinterpret_cast<uint4*>(&tile[row][ch * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = threadIdx.x + i * 256, oc = idx & 63, rch = idx >> 6;
    const int c = c0 + oc, r = r0 + rch * 8;
    if (c < C && r < R_pad) {
      uint32_t w[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[rch * 8 + 2 * e][oc] | ((uint32_t)tile[rch * 8 + 2 * e + 1][oc] << 16);
      *reinterpret_cast<uint4*>(out + (int64_t)c * ld_out + r) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

extern "C" int sf_transpose_bf16_multi(const int64_t* table, const int* tile_prefix, int n_tensors, int total_tiles, void* stream) {
  SF_CHECK_ARG(table && tile_prefix && n_tensors >= 1 && total_tiles >= 1, "sf_transpose_bf16_multi: bad arguments");
  hipLaunchKernelGGL(transpose_bf16_multi_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, table, tile_prefix, n_tensors);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// fp32 -> bf16 cast of a (rows, cols) matrix (cols % 4 == 0), optional scale.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, int64_t ldx, bf16_t* __restrict__ y, int64_t ldy,
                                                         int64_t rows, int cols4, float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols4) return;
  const int64_t r = i / cols4; const int c = (int)(i - r * cols4) * 4;
  const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
  uint2 o; o.x = pack_bf2(v.x * scale, v.y * scale); o.y = pack_bf2(v.z * scale, v.w * scale);
  *reinterpret_cast<uint2*>(y + r * ldy + c) = o;
}

extern "C" int sf_cast_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int64_t rows, int cols, float scale, void* stream) {
  SF_CHECK_ARG(x && y && (cols % 4) == 0 && (ldx % 4) == 0 && (ldy % 4) == 0, "sf_cast_bf16: bad arguments");
  const int64_t n = rows * (cols / 4);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, rows, cols / 4,
                     scale);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Row softmax (attention probabilities recomputed for the backward): P[r, :L] = softmax(S[r, :L]) as bf16, P[r, L:L_pad] = 0.
// One wave per row, L <= 256; `scale` (1/sqrt(d)) is applied to S first.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, int64_t lds_, bf16_t* __restrict__ P, int64_t ldp,
                                                            int64_t rows, int L, int L_pad, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float v[4], m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; v[i] = c < L ? S[r * lds_ + c] * scale : -INFINITY; m = fmaxf(m, v[i]); }
  m = wave_max(m);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = __expf(v[i] - m); s += v[i]; }
  const float inv = 1.0f / wave_sum(s);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; if (c < L_pad) P[r * ldp + c] = c < L ? f2bf(v[i] * inv) : (bf16_t)0; }
}

extern "C" int sf_softmax_rows(const float* S, int64_t lds_, uint16_t* P, int64_t ldp, int64_t rows, int L, int L_pad, float scale, void* stream) {
  SF_CHECK_ARG(S && P && L >= 1 && L <= 256 && L_pad >= L && L_pad <= 256, "sf_softmax_rows: bad arguments");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, lds_, P, ldp, rows, L, L_pad, scale);
  SF_LAUNCH_CHECK();
  return 0;
}

// dS[r, c] = scale * P[r, c] * (dP[r, c] - sum_j P[r, j] dP[r, j])  (bf16, zero-padded to L_pad)
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const bf16_t* __restrict__ P, int64_t ldp, const float* __restrict__ dP,
                                                                int64_t lddp, bf16_t* __restrict__ dS, int64_t ldds, int64_t rows, int L,
                                                                int L_pad, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float p[4], g[4], dot = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = lane + 64 * i;
    p[i] = c < L ? bf2f(P[r * ldp + c]) : 0.f;
    g[i] = c < L ? dP[r * lddp + c] : 0.f;
    dot += p[i] * g[i];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; if (c < L_pad) dS[r * ldds + c] = c < L ? f2bf(scale * p[i] * (g[i] - dot)) : (bf16_t)0; }
}

extern "C" int sf_softmax_bwd_rows(const uint16_t* P, int64_t ldp, const float* dP, int64_t lddp, uint16_t* dS, int64_t ldds, int64_t rows,
                                   int L, int L_pad, float scale, void* stream) {
  SF_CHECK_ARG(P && dP && dS && L >= 1 && L <= 256 && L_pad >= L && L_pad <= 256, "sf_softmax_bwd_rows: bad arguments");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, ldp, dP, lddp, dS, ldds,
                     rows, L, L_pad, scale);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// LayerNorm backward over 768 columns (statistics recomputed from x):
//   dx[omap(r)] (=|+=) rstd * (g*dy - mean(g*dy) - xhat * mean(g*dy*xhat)),  xhat = (x[imap(r)] - mean) * rstd
//   part[blk][0][c] = sum_rows dy*xhat (dgamma), part[blk][1][c] = sum_rows dy (dbeta), reduced by sf_colsum_partials.
// One wave per row, 4 * rpw rows per block (each wave walks rpw rows, keeping its dgamma / dbeta partials in registers) -> the
// block's 4 waves combine their per-column partials through LDS.
// ------------------------------------------------------------------------------------------------------
template <bool DY_BF16>
__global__ __launch_bounds__(256) void layernorm768_bwd_kernel(const float* __restrict__ x, int64_t ldx, RowMap xmap, const float* __restrict__ gamma,
                                                                const void* __restrict__ dy, int64_t lddy, RowMap dymap, float* __restrict__ dx,
                                                                int64_t lddx, RowMap dxmap, int accumulate, float* __restrict__ part, int64_t rows,
                                                                float eps, int rpw, bf16_t* __restrict__ ynext = nullptr, int64_t ldyn = 0,
                                                                const float* __restrict__ seq_scale = nullptr, int64_t seq_rows = 1) {
  // ynext (sf_layernorm768_bwd_branch): the updated dx row is also the input of the NEXT residual branch's backward - written here as that branch's dY operand,
  // bf16(scale[row / seq_rows] * dx), with its fp32 column sums (the branch's output-bias gradient) as a third partial plane: sf_branch_grad's pass over dx saved
  __shared__ float red[4][3][768];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int planes = ynext ? 3 : 2;
  float4 dg[3], db[3], dn[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { dg[i] = make_float4(0.f, 0.f, 0.f, 0.f); db[i] = dg[i]; dn[i] = dg[i]; }
  for (int it = 0; it < rpw; ++it) {
    const int64_t r = ((int64_t)blockIdx.x * rpw + it) * 4 + wave;
    if (r >= rows) break;
    const float* xr = x + map_row(xmap, r) * ldx;
    const int64_t gro = map_row(dymap, r) * lddy;
    float4 v[3], g[3];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v[i] = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
      if (DY_BF16) {                                               // incoming gradient in bf16 (written so by the dgrad GEMM): 8-byte loads
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(dy) + gro + i * 256 + lane * 4);
        g[i] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
      } else {
        g[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(dy) + gro + i * 256 + lane * 4);
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) * (1.0f / 768);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(wave_sum(q) * (1.0f / 768) + eps);
    float a = 0.f, b = 0.f;   // sum(g*dy), sum(g*dy*xhat)
    float4 gd[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 w = *reinterpret_cast<const float4*>(gamma + i * 256 + lane * 4);
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;           // xhat
      gd[i] = make_float4(w.x * g[i].x, w.y * g[i].y, w.z * g[i].z, w.w * g[i].w);
      a += (gd[i].x + gd[i].y) + (gd[i].z + gd[i].w);
      b += (gd[i].x * v[i].x + gd[i].y * v[i].y) + (gd[i].z * v[i].z + gd[i].w * v[i].w);
      dg[i].x += g[i].x * v[i].x; dg[i].y += g[i].y * v[i].y; dg[i].z += g[i].z * v[i].z; dg[i].w += g[i].w * v[i].w;
      db[i].x += g[i].x; db[i].y += g[i].y; db[i].z += g[i].z; db[i].w += g[i].w;
    }
    a = wave_sum(a) * (1.0f / 768); b = wave_sum(b) * (1.0f / 768);
    float* dr = dx + map_row(dxmap, r) * lddx;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float4 o;
      o.x = rstd * (gd[i].x - a - v[i].x * b); o.y = rstd * (gd[i].y - a - v[i].y * b);
      o.z = rstd * (gd[i].z - a - v[i].z * b); o.w = rstd * (gd[i].w - a - v[i].w * b);
      float* dp = dr + i * 256 + lane * 4;
      if (accumulate) { const float4 t = *reinterpret_cast<const float4*>(dp); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
      *reinterpret_cast<float4*>(dp) = o;
      if (ynext) {                                                   // (identity row maps on this path)
        const float sc = seq_scale ? seq_scale[r / seq_rows] : 1.0f;
        o.x *= sc; o.y *= sc; o.z *= sc; o.w *= sc;                  // a dropped branch (sc = 0): zero rows, no bias gradient
        dn[i].x += o.x; dn[i].y += o.y; dn[i].z += o.z; dn[i].w += o.w;
        uint2 w2; w2.x = pack_bf2(o.x, o.y); w2.y = pack_bf2(o.z, o.w);
        *reinterpret_cast<uint2*>(ynext + r * ldyn + i * 256 + lane * 4) = w2;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    *reinterpret_cast<float4*>(&red[wave][0][i * 256 + lane * 4]) = dg[i];
    *reinterpret_cast<float4*>(&red[wave][1][i * 256 + lane * 4]) = db[i];
    if (ynext) *reinterpret_cast<float4*>(&red[wave][2][i * 256 + lane * 4]) = dn[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < planes * 768; c += 256) {
    const int k = c / 768, col = c - k * 768;
    part[((int64_t)blockIdx.x * planes + k) * 768 + col] = (red[0][k][col] + red[1][k][col]) + (red[2][k][col] + red[3][k][col]);
  }
}

// out[c] (=|+=) sum_p part[p * stride + c], c < cols  (second stage of the two-stage column reductions)
// out2 != NULL: columns >= cols_split go to out2[c - cols_split] (the dgamma | dbeta pair of the LayerNorm backward in one launch)
__global__ __launch_bounds__(1024) void colsum_partials_kernel(const float* __restrict__ part, int64_t n_part, int64_t stride, float* __restrict__ out,
                                                                int cols, int accumulate, float* __restrict__ out2 = nullptr, int cols_split = 0) {
  // 64 columns x 16 interleaved row lanes per workgroup, four independent partial sums per thread: enough loads in flight that the
  // second stage of the column reductions is not a chain of dependent L2 round trips (it was 9 % of the Stage-1 train step)
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int64_t p = g;
    for (; p + 48 < n_part; p += 64) {
      s0 += part[p * stride + c];
      s1 += part[(p + 16) * stride + c];
      s2 += part[(p + 32) * stride + c];
      s3 += part[(p + 48) * stride + c];
    }
    for (; p < n_part; p += 16) s0 += part[p * stride + c];
  }
  red[g][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && c < cols) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][cl];
    float* o = (out2 && c >= cols_split) ? out2 + (c - cols_split) : out + c;
    *o = accumulate ? *o + s : s;
  }
}

static int layernorm768_bwd_impl(const float* x, int64_t ldx, const int64_t* x_map, const float* gamma, const void* dy, bool dy_bf16, int64_t lddy,
                                 const int64_t* dy_map, float* dx, int64_t lddx, const int64_t* dx_map, int accumulate_dx, float* dgamma,
                                 float* dbeta, int accumulate_dparams, float* workspace, int64_t rows, float eps, void* stream, bf16_t* ynext = nullptr,
                                 int64_t ldyn = 0, const float* seq_scale = nullptr, int64_t seq_rows = 1, float* dbias_next = nullptr) {
  SF_CHECK_ARG(x && gamma && dy && dx && dgamma && dbeta && workspace, "sf_layernorm768_bwd: null pointer");
  if (rows <= 0) return 0;
  const int planes = ynext ? 3 : 2;
  const int rpw = rows >= 16384 ? 8 : (rows >= 4096 ? 2 : 1);      // keep >= 1k blocks in flight, <= ~1.4k partial rows at Stage-1 sizes
  const int64_t nblk = (rows + 4 * rpw - 1) / (4 * rpw);
  hipStream_t s = (hipStream_t)stream;
  if (dy_bf16) hipLaunchKernelGGL(layernorm768_bwd_kernel<true>, dim3((unsigned)nblk), dim3(256), 0, s, x, ldx, sf_rowmap(x_map), gamma, dy, lddy,
                                  sf_rowmap(dy_map), dx, lddx, sf_rowmap(dx_map), accumulate_dx, workspace, rows, eps, rpw, ynext, ldyn, seq_scale, seq_rows);
  else hipLaunchKernelGGL(layernorm768_bwd_kernel<false>, dim3((unsigned)nblk), dim3(256), 0, s, x, ldx, sf_rowmap(x_map), gamma, dy, lddy,
                          sf_rowmap(dy_map), dx, lddx, sf_rowmap(dx_map), accumulate_dx, workspace, rows, eps, rpw, ynext, ldyn, seq_scale, seq_rows);
  SF_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_partials_kernel, dim3(24), dim3(1024), 0, s, workspace, nblk, (int64_t)planes * 768, dgamma, 2 * 768, accumulate_dparams, dbeta, 768);
  SF_LAUNCH_CHECK();
  if (ynext) {
    hipLaunchKernelGGL(colsum_partials_kernel, dim3(12), dim3(1024), 0, s, workspace + 2 * 768, nblk, (int64_t)planes * 768, dbias_next, 768, 0, (float*)nullptr, 0);
    SF_LAUNCH_CHECK();
  }
  return 0;
}

// sf_layernorm768_bwd(_bf16) with identity row maps that ALSO writes the head of the next residual branch's backward (what sf_branch_grad would compute from the
// updated dx in another pass): y_next = bf16(seq_scale[row / seq_rows] * dx) (seq_scale NULL = 1) and dbias_next[c] = sum_r seq_scale * dx[r, c] (fp32, assigned).
// workspace: fp32, 3 * 768 * ceil(rows / 4) elements.
extern "C" int sf_layernorm768_bwd_branch(const float* x, int64_t ldx, const float* gamma, const void* dy, int dy_dtype, int64_t lddy, float* dx, int64_t lddx,
                                          int accumulate_dx, float* dgamma, float* dbeta, int accumulate_dparams, uint16_t* y_next, int64_t ldyn,
                                          const float* seq_scale, int64_t seq_rows, float* dbias_next, float* workspace, int64_t rows, float eps, void* stream) {
  SF_CHECK_ARG(y_next && dbias_next && (ldyn % 4) == 0 && ((uintptr_t)y_next % 8) == 0 && seq_rows >= 1, "sf_layernorm768_bwd_branch: bad next-branch arguments");
  SF_CHECK_ARG(dy_dtype == SF_BF16 || dy_dtype == SF_F32, "sf_layernorm768_bwd_branch: dy must be bf16 or f32");
  return layernorm768_bwd_impl(x, ldx, nullptr, gamma, dy, dy_dtype == SF_BF16, lddy, nullptr, dx, lddx, nullptr, accumulate_dx, dgamma, dbeta, accumulate_dparams,
                               workspace, rows, eps, stream, y_next, ldyn, seq_scale, seq_rows, dbias_next);
}

extern "C" int sf_layernorm768_bwd(const float* x, int64_t ldx, const int64_t* x_map, const float* gamma, const float* dy, int64_t lddy,
                                   const int64_t* dy_map, float* dx, int64_t lddx, const int64_t* dx_map, int accumulate_dx, float* dgamma,
                                   float* dbeta, int accumulate_dparams, float* workspace, int64_t rows, float eps, void* stream) {
  return layernorm768_bwd_impl(x, ldx, x_map, gamma, dy, false, lddy, dy_map, dx, lddx, dx_map, accumulate_dx, dgamma, dbeta, accumulate_dparams,
                               workspace, rows, eps, stream);
}

extern "C" int sf_layernorm768_bwd_bf16(const float* x, int64_t ldx, const int64_t* x_map, const float* gamma, const uint16_t* dy, int64_t lddy,
                                        const int64_t* dy_map, float* dx, int64_t lddx, const int64_t* dx_map, int accumulate_dx, float* dgamma,
                                        float* dbeta, int accumulate_dparams, float* workspace, int64_t rows, float eps, void* stream) {
  SF_CHECK_ARG((lddy % 4) == 0 && ((uintptr_t)dy % 8) == 0, "sf_layernorm768_bwd_bf16: dy rows must be 8-byte aligned");
  return layernorm768_bwd_impl(x, ldx, x_map, gamma, dy, true, lddy, dy_map, dx, lddx, dx_map, accumulate_dx, dgamma, dbeta, accumulate_dparams,
                               workspace, rows, eps, stream);
}

// ------------------------------------------------------------------------------------------------------
// Column sums (bias gradients) of a (rows, cols) fp32 | bf16 matrix -> fp32 (cols); and per-position sums over
// sequences (positional-table gradient): out[l, c] = sum_b x[(b*L + l), c].
// ------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void colsum_stage1_kernel(const void* __restrict__ x, int64_t ldx, int64_t rows, int cols, int rows_per_blk,
                                                             float* __restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_blk, r1 = min(r0 + rows_per_blk, rows);
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += BF16 ? bf2f(reinterpret_cast<const bf16_t*>(x)[r * ldx + c]) : reinterpret_cast<const float*>(x)[r * ldx + c];
  part[(int64_t)blockIdx.y * cols + c] = s;
}

// Vector variant: a block covers rows_per_blk rows x 8 sixteen-byte column chunks (64 bf16 / 32 fp32 columns); 8 neighbouring lanes read one
// whole 128-byte line of a row, 32 row lanes stride the rows (the scalar kernel above moves 2 or 4 bytes per lane per load).
template <bool BF16>
__global__ __launch_bounds__(256) void colsum_stage1_vec_kernel(const void* __restrict__ x, int64_t ldx, int64_t rows, int cols, int rows_per_blk,
                                                                 float* __restrict__ part) {
  constexpr int E = BF16 ? 8 : 4;                                 // elements per 16-byte chunk
  __shared__ float red[4][8 * 8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = tid >> 3, ch = tid & 7;
  const int c0 = blockIdx.x * (8 * E) + ch * E;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_blk, r1 = min(r0 + rows_per_blk, rows);
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int64_t r = r0 + rl; r < r1; r += 32) {
    if (BF16) {
      const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(x) + r * ldx + c0);
      const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc[2 * i] += __uint_as_float(w[i] << 16); acc[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
    } else {
      const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(x) + r * ldx + c0);
      acc[0] += u.x; acc[1] += u.y; acc[2] += u.z; acc[3] += u.w;
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {                                   // over the 8 row lanes of the wave that share this chunk
    float t = acc[e];
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    if (lane < 8) red[wave][ch * 8 + e] = t;
  }
  __syncthreads();
  if (tid < 8 * E) {
    const int cch = tid / E, e = tid - cch * E;
    part[(int64_t)blockIdx.y * cols + blockIdx.x * (8 * E) + tid] = (red[0][cch * 8 + e] + red[1][cch * 8 + e]) + (red[2][cch * 8 + e] + red[3][cch * 8 + e]);
  }
}

extern "C" int sf_colsum(const void* x, int x_dtype, int64_t ldx, int64_t rows, int cols, float* out, int accumulate, float* workspace,
                         void* stream) {
  SF_CHECK_ARG(x && out && workspace && (x_dtype == SF_F32 || x_dtype == SF_BF16), "sf_colsum: bad arguments");
  if (rows <= 0 || cols <= 0) return 0;
  const int rpb = rows >= 16384 ? 256 : 64;       // workspace contract (cols * ceil(rows / 64)) is an upper bound
  const int64_t nblk = (rows + rpb - 1) / rpb;
  SF_CHECK_ARG(nblk < 65536, "sf_colsum: too many rows");
  dim3 grid((cols + 255) / 256, (unsigned)nblk);
  hipStream_t s = (hipStream_t)stream;
  const int cw = x_dtype == SF_BF16 ? 64 : 32, esz = x_dtype == SF_BF16 ? 2 : 4;
  if ((cols % cw) == 0 && ((uintptr_t)x % 16) == 0 && (ldx * esz) % 16 == 0) {
    dim3 vgrid(cols / cw, (unsigned)nblk);
    if (x_dtype == SF_BF16) hipLaunchKernelGGL((colsum_stage1_vec_kernel<true>), vgrid, dim3(256), 0, s, x, ldx, rows, cols, rpb, workspace);
    else hipLaunchKernelGGL((colsum_stage1_vec_kernel<false>), vgrid, dim3(256), 0, s, x, ldx, rows, cols, rpb, workspace);
  } else if (x_dtype == SF_BF16) hipLaunchKernelGGL((colsum_stage1_kernel<true>), grid, dim3(256), 0, s, x, ldx, rows, cols, rpb, workspace);
  else hipLaunchKernelGGL((colsum_stage1_kernel<false>), grid, dim3(256), 0, s, x, ldx, rows, cols, rpb, workspace);
  SF_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((cols + 63) / 64), dim3(1024), 0, s, workspace, nblk, (int64_t)cols, out, cols, accumulate);
  SF_LAUNCH_CHECK();
  return 0;
}

// out[r] (=|+=) sum_c x[r, c] for a bf16 (rows, cols) matrix, cols % 8 == 0: bias gradients read off the TRANSPOSED gradient copy the
// weight-gradient GEMM needs anyway (row r of dy^T = column r of dy) - contiguous 16-byte reads, one wave per row, no second stage.
__global__ __launch_bounds__(256) void rowsum_bf16_kernel(const bf16_t* __restrict__ x, int64_t ldx, int rows, int64_t cols, float* __restrict__ out,
                                                           int accumulate) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= rows) return;
  const bf16_t* xr = x + (int64_t)r * ldx;
  float s0 = 0.f, s1 = 0.f;
  for (int64_t c = (int64_t)lane * 8; c < cols; c += 512) {
    const uint4 v = *reinterpret_cast<const uint4*>(xr + c);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { s0 += __uint_as_float(w[e] << 16); s1 += __uint_as_float(w[e] & 0xffff0000u); }
  }
  const float s = wave_sum(s0 + s1);
  if (lane == 0) out[r] = accumulate ? out[r] + s : s;
}

extern "C" int sf_rowsum_bf16(const uint16_t* x, int64_t ldx, int rows, int64_t cols, float* out, int accumulate, void* stream) {
  SF_CHECK_ARG(x && out && rows >= 1 && cols >= 8 && (cols % 8) == 0 && (ldx % 8) == 0 && ((uintptr_t)x % 16) == 0, "sf_rowsum_bf16: bad arguments");
  hipLaunchKernelGGL(rowsum_bf16_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, rows, cols, out, accumulate);
  SF_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void seqsum_kernel(const float* __restrict__ x, int64_t ldx, int n_seq, int L, int cols, float* __restrict__ out,
                                                      int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)L * cols) return;
  const int l = (int)(i / cols), c = (int)(i - (int64_t)l * cols);
  float s = 0.f;
  for (int b = 0; b < n_seq; ++b) s += x[((int64_t)b * L + l) * ldx + c];
  out[i] = accumulate ? out[i] + s : s;
}

extern "C" int sf_seqsum(const float* x, int64_t ldx, int n_seq, int L, int cols, float* out, int accumulate, void* stream) {
  SF_CHECK_ARG(x && out && n_seq >= 1 && L >= 1 && cols >= 1, "sf_seqsum: bad arguments");
  const int64_t n = (int64_t)L * cols;
  hipLaunchKernelGGL(seqsum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, n_seq, L, cols, out, accumulate);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Sum of the split-K partials of a weight gradient - and, in the same launch, of its bias gradient:
//     dW[i] = sum_s part[s * n_w + i]  (i < n_w = N * K),      db[j] (=|+=) sum_s bpart[s * n_b + j]  (j < n_b = N; bpart may be NULL)
// 16-byte loads, four chunks in flight per lane (sf_seqsum moved 4 bytes per lane per load: 18-20 us for the 66 MB of a 27-chunk 768 x 768 gradient,
// plus a second ~5 us launch for the bias).  n_w % 4 == 0, n_b % 4 == 0.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wgrad_sum_kernel(const float* __restrict__ part, int64_t n_w4, int split, float* __restrict__ dw,
                                                         const float* __restrict__ bpart, int64_t n_b4, float* __restrict__ db, int acc_bias) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float4* src; float4* dst; int64_t stride4; bool acc = false;
  if (i < n_w4) { src = reinterpret_cast<const float4*>(part) + i; dst = reinterpret_cast<float4*>(dw) + i; stride4 = n_w4; }
  else if (i < n_w4 + n_b4) { const int64_t j = i - n_w4; src = reinterpret_cast<const float4*>(bpart) + j; dst = reinterpret_cast<float4*>(db) + j; stride4 = n_b4; acc = acc_bias != 0; }
  else return;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
  int s_ = 0;
  for (; s_ + 4 <= split; s_ += 4) {
    const float4 v0 = src[(int64_t)s_ * stride4], v1 = src[(int64_t)(s_ + 1) * stride4], v2 = src[(int64_t)(s_ + 2) * stride4], v3 = src[(int64_t)(s_ + 3) * stride4];
    a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
    a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
    a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
    a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
  }
  for (; s_ < split; ++s_) { const float4 v = src[(int64_t)s_ * stride4]; a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w; }
  float4 r = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
  if (acc) { const float4 t = *dst; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
  *dst = r;
}

extern "C" int sf_wgrad_sum(const float* part, int64_t n_w, int split, float* dw, const float* bias_part, int64_t n_b, float* db, int accumulate_bias,
                            void* stream) {
  SF_CHECK_ARG(part && dw && split >= 1 && n_w >= 4 && (n_w % 4) == 0 && ((uintptr_t)part % 16) == 0 && ((uintptr_t)dw % 16) == 0, "sf_wgrad_sum: bad weight arguments");
  SF_CHECK_ARG(!bias_part || (db && n_b >= 4 && (n_b % 4) == 0 && ((uintptr_t)bias_part % 16) == 0 && ((uintptr_t)db % 16) == 0), "sf_wgrad_sum: bad bias arguments");
  const int64_t n4 = n_w / 4 + (bias_part ? n_b / 4 : 0);
  hipLaunchKernelGGL(wgrad_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part, n_w / 4, split, dw, bias_part,
                     bias_part ? n_b / 4 : 0, db, accumulate_bias);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// GELU (exact erf form) forward on a saved pre-activation, and its backward:
//   act = gelu(pre);   dpre = dact * (Phi(pre) + pre * phi(pre)),  Phi = 0.5 (1 + erf(x / sqrt 2)), phi = exp(-x^2/2) / sqrt(2 pi)
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ act, int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const uint2 u = reinterpret_cast<const uint2*>(pre)[i];
  sf_f32x2_t a = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u)}, b = {__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
  gelu_erf4(a, b);
  uint2 o;
  o.x = pack_bf2(a.x, a.y);
  o.y = pack_bf2(b.x, b.y);
  reinterpret_cast<uint2*>(act)[i] = o;
}

__device__ __forceinline__ float gelu_grad(float x) {
  const float phi = 0.3989422804014327f * __expf(-0.5f * x * x);
  const float cdf = erff(x * 0.70710678118654752f) * 0.5f + 0.5f;
  return cdf + x * phi;
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ pre, const float* __restrict__ dact, bf16_t* __restrict__ dpre,
                                                        int64_t n4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const uint2 u = reinterpret_cast<const uint2*>(pre)[i];
  const float4 g = reinterpret_cast<const float4*>(dact)[i];
  uint2 o;
  o.x = pack_bf2(g.x * gelu_grad(__uint_as_float(u.x << 16)), g.y * gelu_grad(__uint_as_float(u.x & 0xffff0000u)));
  o.y = pack_bf2(g.z * gelu_grad(__uint_as_float(u.y << 16)), g.w * gelu_grad(__uint_as_float(u.y & 0xffff0000u)));
  reinterpret_cast<uint2*>(dpre)[i] = o;
}

// same with the incoming gradient in bf16 (the fc2 input gradient written in bf16 by the dgrad GEMM: half the bytes of the fp32 round trip)
__global__ __launch_bounds__(256) void gelu_bwd_bf16_kernel(const bf16_t* __restrict__ pre, const bf16_t* __restrict__ dact, bf16_t* __restrict__ dpre,
                                                             int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const uint4 u = reinterpret_cast<const uint4*>(pre)[i], g = reinterpret_cast<const uint4*>(dact)[i];
  const uint32_t uw[4] = {u.x, u.y, u.z, u.w}, gw[4] = {g.x, g.y, g.z, g.w};
  uint32_t ow[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    ow[e] = pack_bf2(__uint_as_float(gw[e] << 16) * gelu_grad(__uint_as_float(uw[e] << 16)),
                     __uint_as_float(gw[e] & 0xffff0000u) * gelu_grad(__uint_as_float(uw[e] & 0xffff0000u)));
  reinterpret_cast<uint4*>(dpre)[i] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

extern "C" int sf_gelu_bwd_bf16(const uint16_t* pre, const uint16_t* dact, uint16_t* dpre, int64_t n, void* stream) {
  SF_CHECK_ARG(pre && dact && dpre && (n % 8) == 0 && ((uintptr_t)pre % 16) == 0 && ((uintptr_t)dact % 16) == 0 && ((uintptr_t)dpre % 16) == 0,
               "sf_gelu_bwd_bf16: bad arguments (n %% 8 == 0, 16-byte aligned)");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gelu_bwd_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, dact, dpre, n / 8);
  SF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sf_gelu_fwd(const uint16_t* pre, uint16_t* act, int64_t n, void* stream) {
  SF_CHECK_ARG(pre && act && (n % 4) == 0, "sf_gelu_fwd: bad arguments");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, act, n / 4);
  SF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sf_gelu_bwd(const uint16_t* pre, const float* dact, uint16_t* dpre, int64_t n, void* stream) {
  SF_CHECK_ARG(pre && dact && dpre && (n % 4) == 0, "sf_gelu_bwd: bad arguments");
  if (n <= 0) return 0;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pre, dact, dpre, n / 4);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Cross-entropy over (B, C) logits with int64 class targets, mean reduction (F.cross_entropy, sync_model.py:95-96):
//   loss = mean_b (logsumexp(z_b) - z_b[t_b]),   dlogits = (softmax(z) - onehot) * (grad_scale / B)
// One workgroup, one wave per row in turn (B and C are tiny: 16 x 21).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cross_entropy_kernel(const float* __restrict__ z, int64_t ldz, const int64_t* __restrict__ tgt, int B, int C,
                                                            float* __restrict__ loss, float* __restrict__ dz, int64_t lddz, float grad_scale) {
  const int lane = threadIdx.x;
  float total = 0.f;
  for (int b = 0; b < B; ++b) {
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, z[b * ldz + c]);
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += __expf(z[b * ldz + c] - m);
    s = wave_sum(s);
    const int64_t t64 = tgt[b];
    const bool ok = t64 >= 0 && t64 < C;            // a target outside [0, C) (an ignore_index, a class of another grid): NaN loss, zero gradient row -
    const int t = ok ? (int)t64 : 0;                //   never an out-of-bounds read (F.cross_entropy raises; a kernel cannot)
    total += ok ? (m + logf(s)) - z[b * ldz + t] : __builtin_nanf("");
    if (dz)
      for (int c = lane; c < C; c += 64) dz[b * lddz + c] = ok ? (__expf(z[b * ldz + c] - m) / s - (c == t ? 1.f : 0.f)) * (grad_scale / B) : 0.f;
  }
  if (lane == 0) *loss = total / B;
}

extern "C" int sf_cross_entropy(const float* logits, int64_t ld, const int64_t* targets, int B, int C, float* loss, float* dlogits, int64_t ldd,
                                float grad_scale, void* stream) {
  SF_CHECK_ARG(logits && targets && loss && B >= 1 && C >= 1, "sf_cross_entropy: bad arguments");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, logits, ld, targets, B, C, loss, dlogits, ldd, grad_scale);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Stochastic depth (DropPath, vit_helper.py:356,372,375; timm.models.layers.DropPath): a residual branch is kept or dropped per SAMPLE
// (= per 0.64 s segment: the towers see (B*S, tokens, D)), kept branches scaled by 1 / keep_prob:
//     y[r, :] = (residual ? residual[r, :] : 0) + seq_scale[r / seq_rows] * x[r, :]          (fp32, cols % 4 == 0)
// seq_scale holds 0 or 1 / keep_prob per sequence (sf_dropout applied to a vector of ones: the same counter-based stream as the
// element dropout, regenerated from (seed, site) in the backward, where the same kernel scales the incoming gradient).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_seq_add_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ seq_scale, int64_t seq_rows,
                                                             const float* __restrict__ r, int64_t ldr, float* __restrict__ y, int64_t ldy, int64_t rows,
                                                             int cols4) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols4) return;
  const int64_t row = i / cols4;
  const int c = (int)(i - row * cols4) * 4;
  const float sc = seq_scale[row / seq_rows];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sc != 0.f) {                                                  // a dropped branch is not even read
    v = *reinterpret_cast<const float4*>(x + row * ldx + c);
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
  }
  if (r) { const float4 t = *reinterpret_cast<const float4*>(r + row * ldr + c); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
  *reinterpret_cast<float4*>(y + row * ldy + c) = v;
}

extern "C" int sf_scale_seq_add(const float* x, int64_t ldx, const float* seq_scale, int64_t seq_rows, const float* residual, int64_t ldr, float* y,
                                int64_t ldy, int64_t rows, int cols, void* stream) {
  SF_CHECK_ARG(x && seq_scale && y && seq_rows >= 1 && cols >= 4 && (cols % 4) == 0, "sf_scale_seq_add: bad arguments");
  SF_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0 && (!residual || (ldr % 4) == 0), "sf_scale_seq_add: row strides must be multiples of 4 elements");
  if (rows <= 0) return 0;
  const int64_t n = rows * (cols / 4);
  hipLaunchKernelGGL(scale_seq_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, seq_scale, seq_rows, residual, ldr,
                     y, ldy, rows, cols / 4);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Whole-token dropout of the sync transformer's inputs (GlobalTransformer.tok_drop_vis / tok_drop_aud: torch.nn.Dropout1d on (B, S, D) drops whole TOKENS,
// sync_model.py:131-134, 160-161): y[ymap(r), :] (=|+=) row_scale[r] * x[xmap(r), :], fp32, row_scale[r] = 0 or 1 / (1 - p) per token (sf_dropout over a vector of
// ones, as the stochastic-depth scales).  Forward: x = the input LayerNorm's rows (contiguous), y = the token matrix with the segment tokens' row map, accumulate on
// top of the positional table; backward: x = the gradient of the token matrix read through the same map, y = the LayerNorm backward's dY rows.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void scale_rows_map_kernel(const float* __restrict__ x, int64_t ldx, RowMap xmap, const float* __restrict__ row_scale,
                                                              float* __restrict__ y, int64_t ldy, RowMap ymap, int64_t rows, int cols4, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols4) return;
  const int64_t row = i / cols4;
  const int c = (int)(i - row * cols4) * 4;
  const float sc = row_scale[row];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sc != 0.f) {                                                  // a dropped token is not even read
    v = *reinterpret_cast<const float4*>(x + map_row(xmap, row) * ldx + c);
    v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
  }
  float* dst = y + map_row(ymap, row) * ldy + c;
  if (accumulate) { const float4 t = *reinterpret_cast<const float4*>(dst); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
  *reinterpret_cast<float4*>(dst) = v;
}

extern "C" int sf_scale_rows_map(const float* x, int64_t ldx, const int64_t* x_map, const float* row_scale, float* y, int64_t ldy, const int64_t* y_map,
                                 int64_t rows, int cols, int accumulate, void* stream) {
  SF_CHECK_ARG(x && row_scale && y && cols >= 4 && (cols % 4) == 0, "sf_scale_rows_map: bad arguments");
  SF_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "sf_scale_rows_map: rows must be 16-byte aligned");
  SF_CHECK_ARG(rows < ((int64_t)1 << 32), "sf_scale_rows_map: the row maps index 32-bit rows");
  if (rows <= 0) return 0;
  const int64_t n = rows * (cols / 4);
  hipLaunchKernelGGL(scale_rows_map_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, sf_rowmap(x_map), row_scale, y, ldy,
                     sf_rowmap(y_map), rows, cols / 4, accumulate);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Head of a residual branch's backward (Stage-1 towers): from the fp32 gradient of the block output dx
//     y[r, :]   = bf16( s[r / seq_rows] * dx[r, :] )              the dY operand of the proj / fc2 weight- and data-gradient GEMMs
//     dbias[c]  (=|+=) sum_r s[r / seq_rows] * dx[r, c]           in fp32, BEFORE the rounding (biases are cancellation-prone)
// in ONE pass over dx (s = per-segment stochastic-depth scale, or absent).  Replaces sf_scale_seq_add -> sf_cast_bf16 -> sf_colsum's first stage:
// three reads of the fp32 gradient and an fp32 copy of the scaled branch gradient.  A block covers rows_per_blk rows x 32 columns (8 lanes = one
// 128-byte line of a row, 32 row lanes); its column sums go to `part` and are reduced by colsum_partials_kernel (second launch).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void branch_grad_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ seq_scale, int64_t seq_rows,
                                                           bf16_t* __restrict__ y, int64_t ldy, int64_t rows, int cols, int rows_per_blk,
                                                           float* __restrict__ part) {
  __shared__ float red[4][8 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = tid >> 3, ch = tid & 7;
  const int c0 = blockIdx.x * 32 + ch * 4;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_blk, r1 = min(r0 + rows_per_blk, rows);
  // a block's rows touch at most two sequences (host: rows_per_blk <= seq_rows)
  float s0 = 1.f, s1 = 1.f;
  int64_t edge = rows;
  if (seq_scale) {
    const int64_t q = r0 / seq_rows;
    edge = (q + 1) * seq_rows;
    s0 = seq_scale[q];
    s1 = edge < r1 ? seq_scale[q + 1] : 0.f;
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t r = r0 + rl; r < r1; r += 32) {
    const float sc = r < edge ? s0 : s1;
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sc != 0.f) {                                                // a dropped branch is not even read
      u = *reinterpret_cast<const float4*>(x + r * ldx + c0);
      u.x *= sc; u.y *= sc; u.z *= sc; u.w *= sc;
    }
    acc[0] += u.x; acc[1] += u.y; acc[2] += u.z; acc[3] += u.w;
    uint2 o; o.x = pack_bf2(u.x, u.y); o.y = pack_bf2(u.z, u.w);
    *reinterpret_cast<uint2*>(y + r * ldy + c0) = o;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {                                     // over the 8 row lanes of the wave that share this chunk
    float t = acc[e];
    t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
    if (lane < 8) red[wave][ch * 4 + e] = t;
  }
  __syncthreads();
  if (tid < 32) part[(int64_t)blockIdx.y * cols + blockIdx.x * 32 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

extern "C" int sf_branch_grad(const float* dx, int64_t ldx, const float* seq_scale, int64_t seq_rows, uint16_t* y, int64_t ldy, int64_t rows, int cols,
                              float* dbias, int accumulate, float* workspace, void* stream) {
  SF_CHECK_ARG(dx && y && dbias && workspace && cols >= 32 && (cols % 32) == 0, "sf_branch_grad: bad arguments (cols must be a multiple of 32)");
  SF_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0 && ((uintptr_t)dx % 16) == 0 && ((uintptr_t)y % 8) == 0, "sf_branch_grad: rows must be 16-byte (dx) / 8-byte (y) aligned");
  if (rows <= 0) return 0;
  int rpb = rows >= 16384 ? 256 : 64;                               // workspace contract as sf_colsum: cols * ceil(rows / 64) floats
  SF_CHECK_ARG(!seq_scale || seq_rows >= 64, "sf_branch_grad: sequences shorter than a 64-row block are not supported");
  if (seq_scale && rpb > seq_rows) rpb = 64;
  const int64_t nblk = (rows + rpb - 1) / rpb;
  SF_CHECK_ARG(nblk < 65536, "sf_branch_grad: too many rows");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(branch_grad_kernel, dim3(cols / 32, (unsigned)nblk), dim3(256), 0, s, dx, ldx, seq_scale, seq_rows, y, ldy, rows, cols, rpb, workspace);
  SF_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((cols + 63) / 64), dim3(1024), 0, s, workspace, nblk, (int64_t)cols, dbias, cols, accumulate);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Fused gradient clipping + Adam over a flat fp32 parameter buffer (make_backward_and_optim_step, train_utils.py:373-386:
// clip_grad_norm_(params, max_norm) then torch.optim.Adam(betas, eps, weight_decay 0); :217-226).
//   stage 1: sum of squares of the flat gradient (two-stage, deterministic) -> norm_out[0] = ||g||_2
//   stage 2: g *= min(1, max_norm / (norm + 1e-6));  m, v update;  p -= lr * mhat / (sqrt(vhat) + eps);  bf16 copy of p
// The norm stays on the device (no host sync); bias corrections are passed in as scalars.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sumsq_stage1_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ part) {
  // 16-byte loads, two of them in flight per lane and four independent partial sums (one scalar load per iteration into one dependent add ran at 2.4 TB/s on the 859 MB
  // Stage-1 gradient buffer); the element order of the sum is fixed by (grid, n) alone: run-to-run identical
  __shared__ float red[4];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int64_t n4 = (((uintptr_t)g & 15) == 0) ? n >> 2 : 0, stride = (int64_t)gridDim.x * 256;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    const float4 a = g4[i], b = g4[i + stride];
    s0 += a.x * a.x + b.x * b.x; s1 += a.y * a.y + b.y * b.y; s2 += a.z * a.z + b.z * b.z; s3 += a.w * a.w + b.w * b.w;
  }
  if (i < n4) { const float4 a = g4[i]; s0 += a.x * a.x; s1 += a.y * a.y; s2 += a.z * a.z; s3 += a.w * a.w; }
  for (int64_t j = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += stride) { const float v = g[j]; s0 += v * v; }   // tail (or an unaligned buffer)
  float s = (s0 + s1) + (s2 + s3);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_stage2_kernel(const float* __restrict__ part, int n_part, float* __restrict__ norm_out) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n_part; i += 256) s += part[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) norm_out[0] = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
}
__global__ __launch_bounds__(256) void adam_clip_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                         bf16_t* __restrict__ p_bf16, int64_t n, const float* __restrict__ norm, float max_norm,
                                                         float lr, float beta1, float beta2, float eps, float bc1, float bc2) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float coef = 1.0f;
  if (max_norm > 0.f) coef = fminf(1.0f, max_norm / (norm[0] + 1e-6f));
  const float gi = g[i] * coef;
  const float mi = beta1 * m[i] + (1.f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
  m[i] = mi; v[i] = vi;
  const float pi = p[i] - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
  p[i] = pi;
  if (p_bf16) p_bf16[i] = f2bf(pi);
}

extern "C" int sf_grad_norm(const float* g, int64_t n, float* norm_out, float* workspace, void* stream) {
  SF_CHECK_ARG(g && norm_out && workspace && n >= 0, "sf_grad_norm: bad arguments");
  const int nblk = 1024;                                          // = the workspace the header asks for (1024 floats)
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sumsq_stage1_kernel, dim3(nblk), dim3(256), 0, s, g, n, workspace);
  hipLaunchKernelGGL(sumsq_stage2_kernel, dim3(1), dim3(256), 0, s, workspace, nblk, norm_out);
  SF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sf_adam_clip_step(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, int64_t n, const float* norm, float max_norm,
                                 float lr, float beta1, float beta2, float eps, int step, void* stream) {
  SF_CHECK_ARG(p && g && m && v && n >= 0 && step >= 1 && (max_norm <= 0.f || norm), "sf_adam_clip_step: bad arguments");
  if (n == 0) return 0;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_clip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, p_bf16, n, norm, max_norm, lr,
                     beta1, beta2, eps, bc1, bc2);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Dropout with a counter-based mask (no mask tensor is stored: backward regenerates it from (seed, element index)):
//   keep(i) = hash(seed, i) >= p * 2^32;   y[i] = (keep ? x[i] / (1 - p) : 0) (+ r[i])
// Sites of the reference's train mode: embd_pdrop on the assembled sequence (sync_model.py:166), resid_pdrop after the
// attention projection and after the MLP (modules/transformer.py:73,90), attn_pdrop on the attention probabilities (:70).
// The mask stream differs from torch's Philox stream, so parity with the reference under dropout is statistical; with the
// masks read back from this kernel the oracle reproduces loss and gradients exactly (tests/test_train_gpu.py).
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t seed, uint32_t idx) {
  uint32_t h = idx * 0x9E3779B1u ^ seed;
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  h += seed * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

template <bool BF16>
__global__ __launch_bounds__(256) void dropout_kernel(const void* __restrict__ x, int64_t ldx, const float* __restrict__ r, int64_t ldr,
                                                       void* __restrict__ y, int64_t ldy, int64_t rows, int cols, uint32_t thresh, float keep_scale,
                                                       uint32_t seed) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t row = i / cols; const int c = (int)(i - row * cols);
  const bool keep = mix32(seed, (uint32_t)i) >= thresh;
  float v = BF16 ? bf2f(reinterpret_cast<const bf16_t*>(x)[row * ldx + c]) : reinterpret_cast<const float*>(x)[row * ldx + c];
  v = keep ? v * keep_scale : 0.f;
  if (r) v += r[row * ldr + c];
  if (BF16) reinterpret_cast<bf16_t*>(y)[row * ldy + c] = f2bf(v);
  else reinterpret_cast<float*>(y)[row * ldy + c] = v;
}

extern "C" int sf_dropout(const void* x, int dtype, int64_t ldx, const float* residual, int64_t ldr, void* y, int64_t ldy, int64_t rows,
                          int cols, float p, uint32_t seed, void* stream) {
  SF_CHECK_ARG(x && y && (dtype == SF_F32 || dtype == SF_BF16) && p >= 0.f && p < 1.f, "sf_dropout: bad arguments");
  SF_CHECK_ARG(rows * (int64_t)cols < ((int64_t)1 << 32), "sf_dropout: more than 2^32 elements");
  const int64_t n = rows * cols;
  if (n <= 0) return 0;
  const uint32_t thresh = (uint32_t)((double)p * 4294967296.0);
  const float ks = 1.0f / (1.0f - p);
  dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == SF_BF16) hipLaunchKernelGGL((dropout_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, residual, ldr, y, ldy, rows, cols, thresh, ks, seed);
  else hipLaunchKernelGGL((dropout_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, residual, ldr, y, ldy, rows, cols, thresh, ks, seed);
  SF_LAUNCH_CHECK();
  return 0;
}
