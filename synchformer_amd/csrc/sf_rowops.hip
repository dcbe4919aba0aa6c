// Row-wise HBM-bound kernels for the 768-wide token matrices of the Synchformer hot path:
// LayerNorm (wave-per-row reduction), table broadcast, row gather/cast, im2col gathers for the two
// patch-embedding convolutions, and the fused u8 -> normalised-bf16 RGB front-end.
// All of these are bandwidth kernels: 16-byte-per-lane coalesced accesses, no LDS, one wave per row.
#include "sf_common.h"
#include "../../include/synchformer_hip.h"

#define D_MODEL 768

// ------------------------------------------------------------------------------------------------------
// LayerNorm over 768 columns.  One wave per row: lane holds 3 x float4 (cols i*256 + lane*4 .. +3), so
// every wave-instruction touches 1 KiB contiguous.  Two-pass (mean, then centred variance) in registers -
// the same arithmetic order class as torch's fp32 LayerNorm.  Output bf16 (GEMM operand) or fp32, optional
// accumulate (y += LN(x)) used to drop normalised features onto a pre-filled positional table.
// Replaces nn.LayerNorm call sites: vit_helper.py:366-375 (norm1/2/3), motionformer.py:232,
// modeling_ast.py:301,315,535, sync_model.py:157,169, modules/transformer.py:94-95.
// ------------------------------------------------------------------------------------------------------
template <bool OUT_BF16, bool ACCUM>
__global__ __launch_bounds__(256) void layernorm768_kernel(const float* __restrict__ x, int64_t ldx, RowMap in_map,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, void* __restrict__ y,
                                                            int64_t ldy, RowMap out_map, int64_t rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + map_row(in_map, r) * ldx;
  float4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) * (1.0f / D_MODEL);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D_MODEL) + eps);
  const int64_t orow = map_row(out_map, r);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 b = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = v[i].x * rstd * g.x + b.x; o.y = v[i].y * rstd * g.y + b.y;
    o.z = v[i].z * rstd * g.z + b.z; o.w = v[i].w * rstd * g.w + b.w;
    if (OUT_BF16) {
      uint2 p; p.x = pack_bf2(o.x, o.y); p.y = pack_bf2(o.z, o.w);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + orow * ldy + c) = p;
    } else {
      float* yp = reinterpret_cast<float*>(y) + orow * ldy + c;
      if (ACCUM) { const float4 t = *reinterpret_cast<const float4*>(yp); o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w; }
      *reinterpret_cast<float4*>(yp) = o;
    }
  }
}

extern "C" int sf_layernorm768(const float* x, int64_t ldx, const int64_t* in_map, const float* gamma,
                               const float* beta, void* y, int y_dtype, int64_t ldy, const int64_t* out_map,
                               int accumulate, int64_t rows, float eps, void* stream) {
  SF_CHECK_ARG(x && gamma && beta && y, "sf_layernorm768: null pointer");
  SF_CHECK_ARG(y_dtype == SF_BF16 || y_dtype == SF_F32, "sf_layernorm768: y_dtype must be bf16 or f32");
  SF_CHECK_ARG(!(accumulate && y_dtype != SF_F32), "sf_layernorm768: accumulate needs f32 output");
  SF_CHECK_ARG((ldx % 4) == 0 && (ldy % 4) == 0, "sf_layernorm768: ld must be a multiple of 4");
  if (rows <= 0) return 0;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  RowMap im = sf_rowmap(in_map), om = sf_rowmap(out_map);
  if (y_dtype == SF_BF16)
    hipLaunchKernelGGL((layernorm768_kernel<true, false>), grid, block, 0, s, x, ldx, im, gamma, beta, y, ldy, om, rows, eps);
  else if (accumulate)
    hipLaunchKernelGGL((layernorm768_kernel<false, true>), grid, block, 0, s, x, ldx, im, gamma, beta, y, ldy, om, rows, eps);
  else
    hipLaunchKernelGGL((layernorm768_kernel<false, false>), grid, block, 0, s, x, ldx, im, gamma, beta, y, ldy, om, rows, eps);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Residual add with a per-sequence branch scale (stochastic depth), fused with the LayerNorm that reads the sum next (Stage-1 towers, forward):
//     x[r, :] = residual[r, :] + seq_scale[r / seq_rows] * branch[r, :]        (fp32)
//     y[r, :] = bf16( LayerNorm(x[r, :]) * gamma + beta )
// `x = x + drop_path(branch)` followed by the next sub-layer's norm (vit_helper.py:364-376).  Was sf_scale_seq_add -> sf_layernorm768: the second launch
// re-read the sum it had just been written.  One wave per row, as layernorm768_kernel; a dropped branch (scale 0) is not read.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void add_scale_ln768_kernel(const float* __restrict__ br, int64_t ldb, const float* __restrict__ seq_scale, int64_t seq_rows,
                                                               const float* __restrict__ res, int64_t ldr, float* __restrict__ x, int64_t ldx,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, bf16_t* __restrict__ y,
                                                               int64_t ldy, int64_t rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float sc = seq_scale ? seq_scale[r / seq_rows] : 1.f;
  float4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const float4*>(res + r * ldr + i * 256 + lane * 4);
  if (sc != 0.f) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float4 b = *reinterpret_cast<const float4*>(br + r * ldb + i * 256 + lane * 4);
      v[i].x += sc * b.x; v[i].y += sc * b.y; v[i].z += sc * b.z; v[i].w += sc * b.w;
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) *reinterpret_cast<float4*>(x + r * ldx + i * 256 + lane * 4) = v[i];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) * (1.0f / D_MODEL);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = rsqrtf(wave_sum(q) * (1.0f / D_MODEL) + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 b = *reinterpret_cast<const float4*>(beta + c);
    uint2 p;
    p.x = pack_bf2(v[i].x * rstd * g.x + b.x, v[i].y * rstd * g.y + b.y);
    p.y = pack_bf2(v[i].z * rstd * g.z + b.z, v[i].w * rstd * g.w + b.w);
    *reinterpret_cast<uint2*>(y + r * ldy + c) = p;
  }
}

extern "C" int sf_add_scale_ln768(const float* branch, int64_t ldb, const float* seq_scale, int64_t seq_rows, const float* residual, int64_t ldr, float* x,
                                  int64_t ldx, const float* gamma, const float* beta, uint16_t* y, int64_t ldy, int64_t rows, float eps, void* stream) {
  SF_CHECK_ARG(branch && residual && x && gamma && beta && y && (!seq_scale || seq_rows >= 1), "sf_add_scale_ln768: bad arguments");
  SF_CHECK_ARG((ldb % 4) == 0 && (ldr % 4) == 0 && (ldx % 4) == 0 && (ldy % 4) == 0, "sf_add_scale_ln768: row strides must be multiples of 4 elements");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(add_scale_ln768_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, branch, ldb, seq_scale, seq_rows, residual, ldr,
                     x, ldx, gamma, beta, y, ldy, rows, eps);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// LayerNorm(768) whose output leaves as MXFP8 (OCP e4m3 + one E8M0 scale per 32 columns, stage-major scale planes - see sf_quantize_mxfp8): the
// A operand of the next MX GEMM, written directly instead of bf16 + a separate quantisation pass (fp8 towers of the synchronizability fine-tune).
// One wave per row; lane l holds columns i*256 + 4 l .. + 3 (i < 3): a 32-column block is 8 consecutive lanes of one i.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm768_mxfp8_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, uint8_t* __restrict__ q, int64_t ldq,
                                                                  uint8_t* __restrict__ sc, int64_t lds, int64_t rows, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + r * ldx;
  float4 v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = wave_sum(s) * (1.0f / D_MODEL);
  float qq = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    qq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
  const float rstd = rsqrtf(wave_sum(qq) * (1.0f / D_MODEL) + eps);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    const float4 g = *reinterpret_cast<const float4*>(gamma + c);
    const float4 b = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = v[i].x * rstd * g.x + b.x; o.y = v[i].y * rstd * g.y + b.y;
    o.z = v[i].z * rstd * g.z + b.z; o.w = v[i].w * rstd * g.w + b.w;
    // the bf16 rounding the un-fused path applies before quantising is kept: the quantiser's input is the SAME value either way
    const uint32_t p01 = pack_bf2(o.x, o.y), p23 = pack_bf2(o.z, o.w);
    o.x = __uint_as_float(p01 << 16); o.y = __uint_as_float(p01 & 0xffff0000u); o.z = __uint_as_float(p23 << 16); o.w = __uint_as_float(p23 & 0xffff0000u);
    float amax = fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w)));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64)); amax = fmaxf(amax, __shfl_xor(amax, 2, 64)); amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    int be = sf_mx_be(amax);
    be = be < 1 ? 1 : (be > 254 ? 254 : be);
    const float inv = __uint_as_float((uint32_t)(254 - be) << 23);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(o.x * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(o.y * inv, 448.f, -448.f), 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(o.z * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(o.w * inv, 448.f, -448.f), w, true);
    *reinterpret_cast<uint32_t*>(q + r * ldq + c) = (uint32_t)w;
    // scale bytes: block (i*8 + lane/8); the four blocks of a 128-column stage (lanes 0-31 or 32-63 of chunk i) form one dword of plane i*2 + lane/32
    uint32_t word = (uint32_t)be << (((lane >> 3) & 3) * 8);
    word |= __shfl_xor(word, 8, 64);
    word |= __shfl_xor(word, 16, 64);
    if ((lane & 31) == 0) *reinterpret_cast<uint32_t*>(sc + (int64_t)(i * 2 + (lane >> 5)) * lds + r * 4) = word;
  }
}

extern "C" int sf_layernorm768_mxfp8(const float* x, int64_t ldx, const float* gamma, const float* beta, uint8_t* q, int64_t ldq, uint8_t* scales,
                                     int64_t lds, int64_t rows, float eps, void* stream) {
  SF_CHECK_ARG(x && gamma && beta && q && scales, "sf_layernorm768_mxfp8: null pointer");
  SF_CHECK_ARG((ldx % 4) == 0 && (ldq % 16) == 0 && (lds % 4) == 0 && lds >= rows * 4 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)scales % 4) == 0,
               "sf_layernorm768_mxfp8: aligned rows and scale planes of >= rows * 4 bytes are required");
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(layernorm768_mxfp8_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, gamma, beta, q, ldq, scales, lds,
                     rows, eps);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// dst[(seq * dst_seq_rows + l) * ld + :] = table[l, :]  for l < L, seq < n_seq   (fp32, cols = 768)
// Lays the positional table (+ CLS / DISTILL / OFF / MOD token rows, folded in at weight-prep time) under
// every sequence; the patch-embed GEMM / LayerNorm then accumulate onto it.  Replaces the cat/expand/add
// glue at video_model_builder.py:221-254, modeling_ast.py:84-90, sync_model.py:153-165,
// motionformer.py:306-307.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void broadcast_rows768_kernel(float* __restrict__ dst, int64_t ld,
                                                                 int64_t dst_seq_rows, const float* __restrict__ table,
                                                                 int64_t L, int64_t total_rows) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= total_rows) return;
  const int64_t seq = r / L, l = r - seq * L;
  const float* t = table + l * D_MODEL;
  float* d = dst + (seq * dst_seq_rows + l) * ld;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    *reinterpret_cast<float4*>(d + i * 256 + lane * 4) = *reinterpret_cast<const float4*>(t + i * 256 + lane * 4);
}

extern "C" int sf_broadcast_rows768(float* dst, int64_t ld, int64_t dst_seq_rows, const float* table, int64_t L,
                                    int64_t n_seq, void* stream) {
  SF_CHECK_ARG(dst && table, "sf_broadcast_rows768: null pointer");
  SF_CHECK_ARG(L > 0 && dst_seq_rows >= L && (ld % 4) == 0, "sf_broadcast_rows768: bad shape");
  const int64_t total = L * n_seq;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(broadcast_rows768_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     dst, ld, dst_seq_rows, table, L, total);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Row gather + cast: y[r, :] = (bf16|f32) x[map(r), :]   (cols = 768, x fp32)
// ------------------------------------------------------------------------------------------------------
template <bool OUT_BF16>
__global__ __launch_bounds__(256) void gather_rows768_kernel(const float* __restrict__ x, int64_t ldx, RowMap in_map,
                                                              void* __restrict__ y, int64_t ldy, int64_t rows) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + map_row(in_map, r) * ldx;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = i * 256 + lane * 4;
    const float4 v = *reinterpret_cast<const float4*>(xr + c);
    if (OUT_BF16) {
      uint2 p; p.x = pack_bf2(v.x, v.y); p.y = pack_bf2(v.z, v.w);
      *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(y) + r * ldy + c) = p;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(y) + r * ldy + c) = v;
    }
  }
}

extern "C" int sf_gather_rows768(const float* x, int64_t ldx, const int64_t* in_map, void* y, int y_dtype, int64_t ldy,
                                 int64_t rows, void* stream) {
  SF_CHECK_ARG(x && y, "sf_gather_rows768: null pointer");
  SF_CHECK_ARG(y_dtype == SF_BF16 || y_dtype == SF_F32, "sf_gather_rows768: y_dtype must be bf16 or f32");
  if (rows <= 0) return 0;
  dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  RowMap im = sf_rowmap(in_map);
  if (y_dtype == SF_BF16)
    hipLaunchKernelGGL((gather_rows768_kernel<true>), grid, block, 0, (hipStream_t)stream, x, ldx, im, y, ldy, rows);
  else
    hipLaunchKernelGGL((gather_rows768_kernel<false>), grid, block, 0, (hipStream_t)stream, x, ldx, im, y, ldy, rows);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Video patch gather (im2col for Conv3d k = s = (2,16,16), vit_helper.py:436-444) fused with the RGB
// front-end (RGBToHalfToZeroOne + RGBNormalize, dataset/transforms.py:647-669) when the input is uint8.
//   vid : (N, T=16, C=3, H=224, W=224)  - the layout Synchformer.forward receives (sync_model.py:43), i.e.
//         BEFORE extract_vfeats' permute; the permute is folded into the gather.
//   out : bf16 (N*1568, 1536), row = n*1568 + f*196 + h*14 + w, col = ((c*2 + dt)*16 + dh)*16 + dw
// One thread moves one 16-pixel patch row (dw run): 16/32/64 B in, 32 B out; a wave covers 64 consecutive
// (patch-col w, dh) runs so reads walk along W within an image row and writes are 32-B pieces of A rows.
// ------------------------------------------------------------------------------------------------------
template <int DT>  // 0 f32, 1 bf16, 2 f16, 3 u8
__device__ __forceinline__ void load16(const void* p, float* f) {
  if (DT == SF_F32) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) { float4 v = q[i]; f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w; }
  } else if (DT == SF_BF16 || DT == SF_F16) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4 v = q[i];
      uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (DT == SF_BF16) { f[8 * i + 2 * j] = bf2f((bf16_t)(w[j] & 0xffff)); f[8 * i + 2 * j + 1] = bf2f((bf16_t)(w[j] >> 16)); }
        else { f[8 * i + 2 * j] = f16_to_f32((uint16_t)(w[j] & 0xffff)); f[8 * i + 2 * j + 1] = f16_to_f32((uint16_t)(w[j] >> 16)); }
      }
    }
  } else {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        // reference: x.half().div(255.) then .sub(0.5).div(0.5) on half tensors (CPU half ops = fp32 op + one
        // rounding to fp16 each): reproduce exactly those roundings.
        const float a = __half2float(__float2half((float)((w[j] >> (8 * b)) & 0xff) / 255.0f));
        const float cen = __half2float(__float2half(a - 0.5f));
        f[4 * j + b] = cen * 2.0f;   // /0.5 is exact
      }
  }
}

template <int DT>
// tok_rows: rows per segment of `out` - 1568 (patch rows only) or 1569 = TOKEN layout: segment n's patches at rows n*1569 + 1 .., row n*1569 (the CLS
// slot) zeroed by 96 extra runs per segment, so that the patch-embedding GEMM runs with identity row maps on the persistent kernel (the CLS rows multiply zeros)
__global__ __launch_bounds__(256) void im2col_video_kernel(const void* __restrict__ vid, bf16_t* __restrict__ out,
                                                            int64_t total_runs, int n_seg_clip, int64_t clip_frames, int frame0, int seg_stride, int tok_rows) {
  // run index = (((n*8 + f)*2 + dt)*3 + c)*224*14 + (h*16+dh)*14 + w   (walks memory order of `vid` per frame)
  const int64_t run = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (run >= total_runs) {
    const int64_t z = run - total_runs, n_total = total_runs / (16 * 3 * 224 * 14);
    if (tok_rows == 1569 && z < n_total * 96) {
      uint4* dst = reinterpret_cast<uint4*>(out + (z / 96) * 1569 * 1536 + (z % 96) * 16);
      dst[0] = make_uint4(0u, 0u, 0u, 0u); dst[1] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
  const int w = (int)(run % 14);
  int64_t t = run / 14;
  const int y = (int)(t % 224); t /= 224;
  const int c = (int)(t % 3); t /= 3;
  const int dt = (int)(t % 2); t /= 2;
  const int f = (int)(t % 8);
  const int64_t n = t / 8;
  // segment n = (clip, s): its 16 frames start at frame clip * clip_frames + frame0 + s * seg_stride of the frame-major input
  const int64_t clip = n / n_seg_clip, sidx = n - clip * n_seg_clip;
  const int64_t frame = clip * clip_frames + frame0 + sidx * seg_stride + (f * 2 + dt);
  const int64_t src = (((frame * 3 + c) * 224 + y) * 224) + w * 16;
  const int esz = (DT == SF_F32) ? 4 : (DT == SF_U8 ? 1 : 2);
  float v[16];
  load16<DT>(reinterpret_cast<const char*>(vid) + src * esz, v);
  const int h = y >> 4, dh = y & 15;
  const int64_t row = n * tok_rows + (tok_rows - 1568) + f * 196 + h * 14 + w;
  const int col = ((c * 2 + dt) * 16 + dh) * 16;
  uint4 o0, o1;
  o0.x = pack_bf2(v[0], v[1]); o0.y = pack_bf2(v[2], v[3]); o0.z = pack_bf2(v[4], v[5]); o0.w = pack_bf2(v[6], v[7]);
  o1.x = pack_bf2(v[8], v[9]); o1.y = pack_bf2(v[10], v[11]); o1.z = pack_bf2(v[12], v[13]); o1.w = pack_bf2(v[14], v[15]);
  uint4* dst = reinterpret_cast<uint4*>(out + row * 1536 + col);
  dst[0] = o0; dst[1] = o1;
}

static int launch_im2col_video(const void* vid, int dtype, bf16_t* out, int64_t n_seg, int n_seg_clip, int64_t clip_frames, int frame0, int seg_stride,
                               hipStream_t s, int tok_rows = 1568) {
  const int64_t total = n_seg * 16 * 3 * 224 * 14;
  if (total <= 0) return 0;
  dim3 grid((unsigned)((total + (tok_rows == 1569 ? n_seg * 96 : 0) + 255) / 256)), block(256);
  switch (dtype) {
    case SF_F32: hipLaunchKernelGGL((im2col_video_kernel<SF_F32>), grid, block, 0, s, vid, out, total, n_seg_clip, clip_frames, frame0, seg_stride, tok_rows); break;
    case SF_BF16: hipLaunchKernelGGL((im2col_video_kernel<SF_BF16>), grid, block, 0, s, vid, out, total, n_seg_clip, clip_frames, frame0, seg_stride, tok_rows); break;
    case SF_F16: hipLaunchKernelGGL((im2col_video_kernel<SF_F16>), grid, block, 0, s, vid, out, total, n_seg_clip, clip_frames, frame0, seg_stride, tok_rows); break;
    default: hipLaunchKernelGGL((im2col_video_kernel<SF_U8>), grid, block, 0, s, vid, out, total, n_seg_clip, clip_frames, frame0, seg_stride, tok_rows); break;
  }
  return 0;
}

extern "C" int sf_im2col_video(const void* vid, int dtype, bf16_t* out, int64_t n_seg, void* stream) {
  SF_CHECK_ARG(vid && out, "sf_im2col_video: null pointer");
  SF_CHECK_ARG(dtype >= 0 && dtype <= 3, "sf_im2col_video: bad dtype %d", dtype);
  launch_im2col_video(vid, dtype, out, n_seg, 1, 16, 0, 0, (hipStream_t)stream);       // every segment is its own 16-frame "clip"
  SF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sf_im2col_video_clips(const void* vid, int dtype, int64_t n_clips, int64_t clip_frames, int frame0, int seg_stride, int n_seg,
                                     bf16_t* out, void* stream) {
  SF_CHECK_ARG(vid && out, "sf_im2col_video_clips: null pointer");
  SF_CHECK_ARG(dtype >= 0 && dtype <= 3, "sf_im2col_video_clips: bad dtype %d", dtype);
  SF_CHECK_ARG(n_seg >= 1 && frame0 >= 0 && seg_stride >= 0 && frame0 + (int64_t)(n_seg - 1) * seg_stride + 16 <= clip_frames,
               "sf_im2col_video_clips: segments [%d + s*%d, +16) do not fit %lld frames", frame0, seg_stride, (long long)clip_frames);
  launch_im2col_video(vid, dtype, out, n_clips * n_seg, n_seg, clip_frames, frame0, seg_stride, (hipStream_t)stream);
  SF_LAUNCH_CHECK();
  return 0;
}

// The same gather into the TOKEN layout: out bf16 (n_clips * n_seg * 1569, 1536), row 0 of every segment zero (see im2col_video_kernel).  n_seg = 1,
// clip_frames = 16, frame0 = seg_stride = 0 is the plain per-segment input of sf_im2col_video.
extern "C" int sf_im2col_video_tokens(const void* vid, int dtype, int64_t n_clips, int64_t clip_frames, int frame0, int seg_stride, int n_seg, bf16_t* out,
                                      void* stream) {
  SF_CHECK_ARG(vid && out, "sf_im2col_video_tokens: null pointer");
  SF_CHECK_ARG(dtype >= 0 && dtype <= 3, "sf_im2col_video_tokens: bad dtype %d", dtype);
  SF_CHECK_ARG(n_seg >= 1 && frame0 >= 0 && seg_stride >= 0 && frame0 + (int64_t)(n_seg - 1) * seg_stride + 16 <= clip_frames,
               "sf_im2col_video_tokens: segments [%d + s*%d, +16) do not fit %lld frames", frame0, seg_stride, (long long)clip_frames);
  launch_im2col_video(vid, dtype, out, n_clips * n_seg, n_seg, clip_frames, frame0, seg_stride, (hipStream_t)stream, 1569);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Spectrogram patch gather (im2col for Conv2d(1->768, k16, stride 10), modeling_ast.py:113-117).
//   spec: fp32 (N, F=128, Ta=66) - the layout Synchformer.forward receives (B,S,1,F,Ta); the reference's
//         permute to (Ta,F) and transpose back (sync_model.py:84, modeling_ast.py:114-115) cancel out.
//   out : bf16 (N*72, 256), row = n*72 + fi*6 + ti, col = df*16 + dt  (value spec[n, 10fi+df, 10ti+dt])
// Tiny (18 K elements per segment): one thread per output element pair.
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_spec_kernel(const float* __restrict__ spec, bf16_t* __restrict__ out,
                                                           int64_t total_pairs, int F, int Ta, int nf, int nt) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= total_pairs) return;
  const int col = (int)(p % 128) * 2;
  const int64_t row = p / 128;
  const int ti = (int)(row % nt);
  const int fi = (int)((row / nt) % nf);
  const int64_t n = row / ((int64_t)nt * nf);
  const int df = col >> 4, dtc = col & 15;
  const float* s = spec + (n * F + (fi * 10 + df)) * Ta + ti * 10 + dtc;
  *reinterpret_cast<uint32_t*>(out + row * 256 + col) = pack_bf2(s[0], s[1]);
}

extern "C" int sf_im2col_spec(const float* spec, bf16_t* out, int64_t n_seg, int F, int Ta, void* stream) {
  SF_CHECK_ARG(spec && out, "sf_im2col_spec: null pointer");
  SF_CHECK_ARG(F >= 16 && Ta >= 16, "sf_im2col_spec: spectrogram smaller than one patch");
  const int nf = (F - 16) / 10 + 1, nt = (Ta - 16) / 10 + 1;
  const int64_t total = n_seg * nf * nt * 128;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(im2col_spec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, spec,
                     out, total, F, Ta, nf, nt);
  SF_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Token masks from content masks: the reference pushes an indicator (1 = kept, inf = masked content) through the patch embedding and
// masks a token iff output channel 0 is NaN (video_model_builder.py:185-201, modeling_ast.py:515-530), i.e. iff the masked elements
// of its patch meet filter-0 weights of BOTH signs (inf - inf) or a zero weight (inf * 0).  w0_sign[k] in {+1, -1, 0} is the sign of
// filter 0 at patch element k (same K order as the im2col gathers).  One wave per token; tok_keep[n*L + t] = 1 keeps the token;
// the CLS (and DISTILL) rows are always kept (video_model_builder.py:223-225).
// ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void token_mask_video_kernel(const uint8_t* __restrict__ keep, const int8_t* __restrict__ w0_sign,
                                                                uint8_t* __restrict__ tok_keep, int64_t n_tok_total) {
  const int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= n_tok_total) return;
  const int64_t n = t / 1569;
  const int tok = (int)(t - n * 1569);
  if (tok == 0) { if (lane == 0) tok_keep[t] = 1; return; }
  const int f = (tok - 1) / 196, hw = (tok - 1) % 196, h = hw / 14, w = hw % 14;
  int flags = 0;                                                   // bit0: +, bit1: -, bit2: zero weight among the masked elements
  for (int e = lane; e < 1536 / 16; e += 64) {                    // e = ((c*2 + dt)*16 + dh): one 16-pixel run each
    const int dh = e & 15, dt = (e >> 4) & 1, c = e >> 5;
    const uint8_t* src = keep + ((((n * 16 + (f * 2 + dt)) * 3 + c) * 224 + (h * 16 + dh)) * 224) + w * 16;
    const uint4 m = *reinterpret_cast<const uint4*>(src);
    const uint32_t mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (((mw[i >> 2] >> (8 * (i & 3))) & 0xffu) == 0) {
        const int sgn = w0_sign[e * 16 + i];
        flags |= sgn > 0 ? 1 : (sgn < 0 ? 2 : 4);
      }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) flags |= __shfl_xor(flags, off, 64);
  if (lane == 0) tok_keep[t] = (((flags & 3) == 3) || (flags & 4)) ? 0 : 1;
}

extern "C" int sf_token_mask_video(const uint8_t* content_keep, int64_t n_seg, const int8_t* w0_sign, uint8_t* tok_keep, void* stream) {
  SF_CHECK_ARG(content_keep && w0_sign && tok_keep && ((uintptr_t)content_keep % 16) == 0, "sf_token_mask_video: bad arguments");
  if (n_seg <= 0) return 0;
  const int64_t total = n_seg * 1569;
  hipLaunchKernelGGL(token_mask_video_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, content_keep, w0_sign, tok_keep,
                     total);
  SF_LAUNCH_CHECK();
  return 0;
}

__global__ __launch_bounds__(256) void token_mask_spec_kernel(const uint8_t* __restrict__ keep, const int8_t* __restrict__ w0_sign,
                                                               uint8_t* __restrict__ tok_keep, int64_t n_tok_total, int F, int Ta, int nf, int nt) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;      // tiny: one thread per token
  if (t >= n_tok_total) return;
  const int L = nf * nt + 2;
  const int64_t n = t / L;
  const int tok = (int)(t - n * L);
  if (tok < 2) { tok_keep[t] = 1; return; }
  const int fi = (tok - 2) / nt, ti = (tok - 2) % nt;
  int flags = 0;
  for (int df = 0; df < 16; ++df)
    for (int dtc = 0; dtc < 16; ++dtc)
      if (keep[(n * F + fi * 10 + df) * Ta + ti * 10 + dtc] == 0) {
        const int sgn = w0_sign[df * 16 + dtc];
        flags |= sgn > 0 ? 1 : (sgn < 0 ? 2 : 4);
      }
  tok_keep[t] = (((flags & 3) == 3) || (flags & 4)) ? 0 : 1;
}

extern "C" int sf_token_mask_spec(const uint8_t* content_keep, int64_t n_seg, int F, int Ta, const int8_t* w0_sign, uint8_t* tok_keep, void* stream) {
  SF_CHECK_ARG(content_keep && w0_sign && tok_keep && F >= 16 && Ta >= 16, "sf_token_mask_spec: bad arguments");
  if (n_seg <= 0) return 0;
  const int nf = (F - 16) / 10 + 1, nt = (Ta - 16) / 10 + 1;
  const int64_t total = n_seg * (nf * nt + 2);
  hipLaunchKernelGGL(token_mask_spec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, content_keep, w0_sign,
                     tok_keep, total, F, Ta, nf, nt);
  SF_LAUNCH_CHECK();
  return 0;
}
