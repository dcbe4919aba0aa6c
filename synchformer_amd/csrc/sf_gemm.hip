// bf16 GEMM with fused epilogues for the Synchformer hot path on gfx950 (MI355X):
//     C[map(m), n] = epi( sum_k A[m, k] * W[n, k] + bias[n] ) (+ R[rmap(m), n])
// A (activations, M x K) and W (nn.Linear weight, N x K) are both K-contiguous, so both MFMA operands are
// "row, 8 consecutive k" fragments and neither needs a transpose.  This one kernel carries >93 % of the
// model's FLOPs: the fused qkv / proj / fc1 / fc2 Linears of every DividedSpaceTimeBlock
// (vit_helper.py:103,155,392-396), ASTLayer (modeling_ast.py:142-146,199,263,274), sync Block
// (modules/transformer.py:59-61,74,86-91), the aggregator layers, the two patch-embedding convolutions
// (as patch-gather GEMMs), vproj/aproj and the offset head.
//
// Tile: 128 x 128 x 64, 256 threads = 4 waves in a 2 x 2 grid, each wave 64 x 64 = 4 x 4 fragments of
// v_mfma_f32_16x16x32_bf16.  Operand tiles go HBM -> LDS directly (global_load_lds_dwordx4, 1 KiB per
// wave-instruction = 8 rows x 128 B), double-buffered, one barrier per K-step.  LDS rows are 128 B; the
// 16-byte chunk index is XOR-swizzled with (row & 7) so the ds_read_b128 fragment reads are conflict-free.
// global_load_lds writes lane-linear, so the swizzle is applied to the per-lane SOURCE address and to the
// read address (same involution), never to the LDS destination.
// Epilogue: accumulators are transposed through (per-wave private) LDS so that bias / GELU / fp32 residual /
// stores run on 16-byte row-contiguous pieces.
// Workgroup -> tile mapping is XCD-aware: the 8 XCDs each walk a contiguous range of tiles, N fastest, so
// an A row-panel is fetched from HBM once per XCD-local L2 and W stays L2/MALL-resident.
#include "sf_common.h"
#include "../../include/synchformer_hip.h"

#define BM 128
#define BN 128
#define BK 64
#define STAGE_BYTES (2 * BM * BK * 2)     // A tile + B tile, bf16
#define EPI_LD 68                          // fp32 row stride of the epilogue staging tile (16-B aligned, padded)

struct GemmArgs {
  const bf16_t* A; int64_t lda;
  const bf16_t* W; int64_t ldw;
  const float* bias;
  void* C; int64_t ldc;
  const float* R; int64_t ldr;
  RowMap cmap, rmap;
  int64_t M;
  int N, K;
  uint32_t tiles_n, tiles_total;
};

__device__ __forceinline__ void glds16(const void* gsrc, char* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

template <bool OUT_BF16, bool GELU, bool HAS_RES>
__global__ __launch_bounds__(256, 2) void gemm_bf16_128x128_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- XCD-aware, bijective block -> tile remap (block b runs on XCD b % 8) -------------------------------
  uint32_t vb;
  {
    const uint32_t nb = p.tiles_total, q = nb >> 3, r = nb & 7u, xcd = blockIdx.x & 7u, idx = blockIdx.x >> 3;
    vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const uint32_t tm = vb / p.tiles_n, tn = vb - tm * p.tiles_n;
  const int64_t m0 = (int64_t)tm * BM;
  const int n0 = (int)tn * BN;

  // ---- per-lane source pointers for the 4 + 4 LDS-DMA pieces this wave issues per stage -------------------
  const int piece_row = lane >> 3;                              // row inside the 8-row piece
  const int gchunk = (lane & 7) ^ piece_row;                    // source-side swizzle (row & 7 == piece_row)
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + piece_row;
    int64_t ar = m0 + row; if (ar > p.M - 1) ar = p.M - 1;      // clamp: tail rows re-read the last valid row
    int br = n0 + row; if (br > p.N - 1) br = p.N - 1;
    a_src[i] = p.A + ar * p.lda + gchunk * 8;
    b_src[i] = p.W + (int64_t)br * p.ldw + gchunk * 8;
  }
  auto stage = [&](int s, int kt) {
    char* base = smem + s * STAGE_BYTES + (wave * 4) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + kt * BK, base + i * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(b_src[i] + kt * BK, base + BM * BK * 2 + i * 1024);
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  // byte offset of this lane's fragment row inside a tile + swizzled chunk offsets for k-step 0 / 1
  const int a_row_off = (wm * 64 + fr) * 128, b_row_off = (wn * 64 + fr) * 128;
  const int ch0 = ((fg) ^ (fr & 7)) * 16, ch1 = ((4 + fg) ^ (fr & 7)) * 16;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
    const char* sa = smem + (kt & 1) * STAGE_BYTES;
    const char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ks ? ch1 : ch0;
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sa + a_row_off + i * 16 * 128 + ch);
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb + b_row_off + j * 16 * 128 + ch);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  __syncthreads();   // every wave is done reading operand tiles; LDS becomes per-wave epilogue scratch

  // ---- epilogue: 2 passes of 32 rows x 64 cols through this wave's private LDS slab -----------------------
  float* slab = reinterpret_cast<float*>(smem + wave * (32 * EPI_LD * 4));
  const int ecol = (lane & 15) * 4;                 // 4 consecutive output columns per lane
  const int gcol = n0 + wn * 64 + ecol;
  const bool vec_ok = ((p.N & 3) == 0);
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) {
    if (vec_ok && gcol + 3 < p.N) bias4 = *reinterpret_cast<const float4*>(p.bias + gcol);
    else {
      if (gcol + 0 < p.N) bias4.x = p.bias[gcol + 0];
      if (gcol + 1 < p.N) bias4.y = p.bias[gcol + 1];
      if (gcol + 2 < p.N) bias4.z = p.bias[gcol + 2];
      if (gcol + 3 < p.N) bias4.w = p.bias[gcol + 3];
    }
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(ii * 16 + fg * 4 + r) * EPI_LD + j * 16 + fr] = acc[half * 2 + ii][j][r];
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int lrow = pass * 4 + (lane >> 4);
      float4 v = *reinterpret_cast<const float4*>(slab + lrow * EPI_LD + ecol);
      const int64_t grow = m0 + wm * 64 + half * 32 + lrow;
      if (grow >= p.M || gcol >= p.N) continue;
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
      const bool full = vec_ok && (gcol + 3 < p.N);
      if (HAS_RES) {
        const float* rp = p.R + map_row(p.rmap, grow) * p.ldr + gcol;
        if (full) { const float4 t = *reinterpret_cast<const float4*>(rp); v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w; }
        else {
          v.x += rp[0];
          if (gcol + 1 < p.N) v.y += rp[1];
          if (gcol + 2 < p.N) v.z += rp[2];
          if (gcol + 3 < p.N) v.w += rp[3];
        }
      }
      const int64_t crow = map_row(p.cmap, grow);
      if (OUT_BF16) {
        bf16_t* cp = reinterpret_cast<bf16_t*>(p.C) + crow * p.ldc + gcol;
        if (full) { uint2 o; o.x = pack_bf2(v.x, v.y); o.y = pack_bf2(v.z, v.w); *reinterpret_cast<uint2*>(cp) = o; }
        else {
          cp[0] = f2bf(v.x);
          if (gcol + 1 < p.N) cp[1] = f2bf(v.y);
          if (gcol + 2 < p.N) cp[2] = f2bf(v.z);
          if (gcol + 3 < p.N) cp[3] = f2bf(v.w);
        }
      } else {
        float* cp = reinterpret_cast<float*>(p.C) + crow * p.ldc + gcol;
        if (full) *reinterpret_cast<float4*>(cp) = v;
        else {
          cp[0] = v.x;
          if (gcol + 1 < p.N) cp[1] = v.y;
          if (gcol + 2 < p.N) cp[2] = v.z;
          if (gcol + 3 < p.N) cp[3] = v.w;
        }
      }
    }
  }
}

template <bool OUT_BF16, bool GELU, bool HAS_RES>
static int launch_gemm(const GemmArgs& a, hipStream_t s) {
  auto kern = gemm_bf16_128x128_kernel<OUT_BF16, GELU, HAS_RES>;
  static bool attr_set = false;   // benign race: the attribute call is idempotent
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES);
    if (e != hipSuccess) { sf_set_error("sf_gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e)); return (int)e; }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(a.tiles_total), dim3(256), 2 * STAGE_BYTES, s, a);
  SF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sf_gemm_bf16(const bf16_t* A, int64_t lda, const bf16_t* W, int64_t ldw, const float* bias, void* C,
                            int c_dtype, int64_t ldc, const int64_t* c_map, const float* R, int64_t ldr,
                            const int64_t* r_map, int epilogue, int64_t M, int64_t N, int64_t K, void* stream) {
  SF_CHECK_ARG(A && W && C, "sf_gemm_bf16: null pointer");
  SF_CHECK_ARG(c_dtype == SF_BF16 || c_dtype == SF_F32, "sf_gemm_bf16: c_dtype must be bf16 or f32");
  SF_CHECK_ARG(epilogue == SF_EPI_NONE || epilogue == SF_EPI_GELU, "sf_gemm_bf16: bad epilogue %d", epilogue);
  SF_CHECK_ARG(K > 0 && (K % BK) == 0, "sf_gemm_bf16: K=%lld must be a positive multiple of %d", (long long)K, BK);
  SF_CHECK_ARG((lda % 8) == 0 && (ldw % 8) == 0, "sf_gemm_bf16: lda/ldw must be multiples of 8 elements (16 B)");
  SF_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0, "sf_gemm_bf16: A/W must be 16-byte aligned");
  SF_CHECK_ARG(M < ((int64_t)1 << 31) && N < ((int64_t)1 << 31), "sf_gemm_bf16: M, N must be < 2^31");
  SF_CHECK_ARG(!(R && c_dtype == SF_BF16 && 0), "unreachable");
  if (M <= 0 || N <= 0) return 0;
  if ((N % 4) == 0) {
    SF_CHECK_ARG((ldc % 4) == 0 && (!R || (ldr % 4) == 0), "sf_gemm_bf16: ldc/ldr must be multiples of 4 when N %% 4 == 0");
  }
  GemmArgs a;
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.bias = bias; a.C = C; a.ldc = ldc; a.R = R; a.ldr = ldr;
  a.cmap = sf_rowmap(c_map); a.rmap = sf_rowmap(r_map);
  a.M = M; a.N = (int)N; a.K = (int)K;
  const int64_t tiles_m = (M + BM - 1) / BM;
  a.tiles_n = (uint32_t)((N + BN - 1) / BN);
  const int64_t total = tiles_m * a.tiles_n;
  SF_CHECK_ARG(total < ((int64_t)1 << 31), "sf_gemm_bf16: too many tiles");
  a.tiles_total = (uint32_t)total;
  hipStream_t s = (hipStream_t)stream;
  const bool gelu = epilogue == SF_EPI_GELU, res = R != nullptr;
  if (c_dtype == SF_BF16) {
    if (gelu) return res ? launch_gemm<true, true, true>(a, s) : launch_gemm<true, true, false>(a, s);
    return res ? launch_gemm<true, false, true>(a, s) : launch_gemm<true, false, false>(a, s);
  }
  if (gelu) return res ? launch_gemm<false, true, true>(a, s) : launch_gemm<false, true, false>(a, s);
  return res ? launch_gemm<false, false, true>(a, s) : launch_gemm<false, false, false>(a, s);
}
