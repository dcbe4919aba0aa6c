// MX-FP8 (OCP e4m3 elements, one E8M0 scale per 32 consecutive k) GEMM with the fused epilogues of sf_gemm_bf16, for gfx950 (MI355X):
//     C[m, n] = epi( sum_k dq(A[m, k]) * dq(W[n, k]) + bias[n] ) (+ R[m, n]),   dq(x[r, k]) = e4m3(x[r, k]) * 2^(scale[r, k / 32] - 127)
// The frozen feature extractors of the synchronizability fine-tune (BASELINE configs[4]; configs/ft_synchability.yaml:7,19 is_trainable False)
// run their four big Linears per block (vit_helper.py:103,155,392-396) on it.  Why block-scaled and not per-tensor fp8: on gfx950 the plain
// fp8 MFMA (v_mfma_f32_32x32x16_fp8_fp8) issues at the bf16 rate; only v_mfma_scale_f32_32x32x64_f8f6f4 - 64-deep, the dequantisation by the two
// block scales fused into the instruction - runs at twice it (MI355X_MICROARCH.md, MFMA table).  And since the bf16 kernels of this repo are bound
// by operand delivery per CU (profiles/r02_gemm_ln.md), halving the operand bytes per k is worth as much as the doubled matrix rate.
//
// Structure = gemm_bf16_persistent_kernel with the same BYTE geometry: a stage is 256 rows x 128 BYTES of A and of W (= 128 k instead of 64), LDS rows
// of 128 B with the 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7), two ring slots, LDS-DMA from inline asm, one counted wait + raw barrier
// per stage, 8 waves as 2 x 4 with 128 x 64 wave tiles.  Per 64-deep MFMA a lane supplies 16 bytes of each of the two MX blocks of its row (the other
// half-wave supplies the other 16) and ONE scale byte - that of block (lane >> 5) - taken from a dword (the 4 block scales of the row for the 128-deep
// stage) that is loaded from the stage-major scale planes one stage ahead.  Which byte of a block lands in which operand dword cannot matter: A and W
// use the same map and the dot product is a sum over the block; which BLOCK a byte is counted in does, and was established on the hardware.
#include "sf_common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/synchformer_hip.h"

#ifndef SF_MX_ABL
#define SF_MX_ABL 0   // measurement builds only (tools/ab_gemm_flags.sh): 2 = no operand refills inside a tile, 4 = no MFMAs
#endif
#ifndef SF_MX_DMA
#define SF_MX_DMA 3   // where the refill's LDS-DMA issues sit (measured at 208 segments, qkv / fc2 us): 0 = A behind step 0, B behind step 1: 1041 / 1064;
#endif                 // 1 = all 8 behind step 0: 1047 / 1056; 2 = A, B behind the two halves of step 0: 981 / 1022; 3 = pairs behind every MFMA pair of step 0: 979 / 993
#define MXBM 256
#define MXBN 256
#define MXBK 128                           // bytes = fp8 elements per row per stage
#define MX_STAGE (2 * MXBM * MXBK)         // 64 KiB: A tile + W tile
#define MX_EPI_LD 64
#define MX_SLAB_BYTES (16 * MX_EPI_LD * 4) // 4 KiB per wave
#define MX_SCALE_OFF (2 * MX_STAGE)          // 2 slots x (1 KiB of A scale dwords + 1 KiB of W scale dwords) behind the operand slots
#define MX_LDS (MX_SCALE_OFF + 4096)       // 132 KiB; the epilogue slabs overlay operand slot 1 (only stage 0 of the next tile is prefetched under the epilogue)
#ifndef SF_MX_STORE_AUX
#define SF_MX_STORE_AUX 2
#endif

typedef __attribute__((ext_vector_type(8))) int mx_i32x8;
typedef __attribute__((ext_vector_type(4))) int mx_i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int mx_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int mx_u32x2;

struct MxArgs {
  const uint8_t* A; int64_t lda; const uint8_t* sA; int64_t ldsa;
  const uint8_t* W; int64_t ldw; const uint8_t* sW; int64_t ldsw;
  const float* bias;
  void* C; int64_t ldc;
  uint8_t* sC; int64_t ldsc;                                     // OUT_FP8: stage-major scale planes of the (quantised) output
  const float* R; int64_t ldr;
  int64_t M;
  int N, K;
  uint32_t tiles_n, tiles_total, nchunk;
};

__device__ __forceinline__ uint32_t mx_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }

// Four LDS-DMA pieces (1 KiB each, consecutive in LDS from the wave-uniform address l0): SGPR base + zero-extended 32-bit lane offsets (the
// operands stay below 4 GiB; 64-bit lane pointers do not fit next to 128 accumulators + the wider MX fragments: they spilled into the k-loop).
__device__ __forceinline__ void mx_dma4(uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3, const void* sbase, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %5\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(l0)
      : "memory", "scc");
}

__device__ __forceinline__ void mx_dma1(uint32_t v0, const void* sbase, uint32_t l0) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "s"(sbase), "s"(l0) : "memory");
}
__device__ __forceinline__ void mx_dma2(uint32_t v0, uint32_t v1, const void* sbase, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(v0), "v"(v1), "s"(sbase), "s"(l0)
      : "memory", "scc");
}

__device__ __forceinline__ void mx_wait_vmcnt0_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((0 & 0xF) | (0x7 << 4) | (0xF << 8) | (0 << 14));
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// OUT: 0 = fp32, 1 = bf16, 2 = MXFP8 (the output is the next MX GEMM's A operand: e4m3 bytes + one E8M0 byte per 32 columns, quantised from the bf16-
// rounded value exactly as sf_quantize_mxfp8 would from a bf16 buffer; a 32-column block = 8 consecutive lanes of the 16-lane row group).
template <int OUT, bool GELU, bool HAS_RES, int NPS = 4>
__device__ __forceinline__ void mx_epi_store(float4 (&v)[NPS], const float4& bias4, const float4 (&res)[NPS], __amdgpu_buffer_rsrc_t rc, uint32_t coff, uint32_t cstep,
                                             __amdgpu_buffer_rsrc_t rsc = __amdgpu_buffer_rsrc_t(), uint32_t sc_off = 0, int rows_left = 0) {
  constexpr bool OUT_BF16 = OUT == 1;
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    float4 x = v[ps];
    x.x += bias4.x; x.y += bias4.y; x.z += bias4.z; x.w += bias4.w;
    if (GELU) {
      sf_f32x2_t g0 = {x.x, x.y}, g1 = {x.z, x.w};
      gelu_erf4(g0, g1);
      x.x = g0.x; x.y = g0.y; x.z = g1.x; x.w = g1.y;
    }
    if (HAS_RES) { x.x += res[ps].x; x.y += res[ps].y; x.z += res[ps].z; x.w += res[ps].w; }
    if (OUT == 2) {
      const uint32_t p01 = pack_bf2(x.x, x.y), p23 = pack_bf2(x.z, x.w);
      const float f0 = __uint_as_float(p01 << 16), f1 = __uint_as_float(p01 & 0xffff0000u), f2 = __uint_as_float(p23 << 16), f3 = __uint_as_float(p23 & 0xffff0000u);
      float amax = fmaxf(fmaxf(fabsf(f0), fabsf(f1)), fmaxf(fabsf(f2), fabsf(f3)));
      // max over the block's eight lanes on DPP (no LDS traffic): quad_perm [1,0,3,2], quad_perm [2,3,0,1], then row_half_mirror (lane i <-> 7 - i)
      amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0xB1, 0xF, 0xF, true)));
      amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0x4E, 0xF, 0xF, true)));
      amax = fmaxf(amax, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(amax), 0x141, 0xF, 0xF, true)));
      int be = sf_mx_be(amax);
      be = be < 1 ? 1 : (be > 254 ? 254 : be);
      const float inv = __uint_as_float((uint32_t)(254 - be) << 23);
      int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f0 * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(f1 * inv, 448.f, -448.f), 0, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f2 * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(f3 * inv, 448.f, -448.f), w, true);
      __builtin_amdgcn_raw_buffer_store_b32((uint32_t)w, rc, coff + ps * cstep, 0, SF_MX_STORE_AUX);
      // one scale byte per row and 32-column block, from the first lane of the block's eight; every other lane (and rows >= M, which would land in the
      // next plane) gets an out-of-range offset that the buffer range check drops - no exec-mask branch in the store loop
      __builtin_amdgcn_raw_buffer_store_b8((uint8_t)be, rsc, ps * 4 < rows_left ? sc_off + ps * 16 : 0xffffffffu, 0, 0);
    } else if (OUT_BF16) {
      mx_u32x2 o; o.x = pack_bf2(x.x, x.y); o.y = pack_bf2(x.z, x.w);
      __builtin_amdgcn_raw_buffer_store_b64(o, rc, coff + ps * cstep, 0, SF_MX_STORE_AUX);
    } else {
      mx_u32x4 o;
      o.x = __float_as_uint(x.x); o.y = __float_as_uint(x.y); o.z = __float_as_uint(x.z); o.w = __float_as_uint(x.w);
      __builtin_amdgcn_raw_buffer_store_b128(o, rc, coff + ps * cstep, 0, SF_MX_STORE_AUX);
    }
  }
}
template <int NPS>
__device__ __forceinline__ void mx_load_res(float4 (&res)[NPS], __amdgpu_buffer_rsrc_t rr, uint32_t roff, uint32_t rstep) {
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    const mx_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rr, roff + ps * rstep, 0, 2);
    res[ps] = make_float4(__uint_as_float(t.x), __uint_as_float(t.y), __uint_as_float(t.z), __uint_as_float(t.w));
  }
}

template <int OUT, bool GELU, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void gemm_mxfp8_persistent_kernel(MxArgs p) {
  constexpr bool OUT_BF16 = OUT == 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;

  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t tiles_m = p.tiles_total / p.tiles_n;
  const uint32_t mp8 = (tiles_m + 7u) >> 3;
  const uint32_t mp0 = min(xcd * mp8, tiles_m), mp1 = min(mp0 + mp8, tiles_m), n_mp = mp1 - mp0;
  const uint32_t gchunk = p.nchunk ? min(p.nchunk, p.tiles_n) : p.tiles_n;
  const uint32_t n_chunks = (p.tiles_n + gchunk - 1) / gchunk, chunk_tiles = n_mp * gchunk;
  const uint32_t t_end = n_mp * p.tiles_n;

  const int piece_row = lane >> 3, slot = lane & 7;
  const int sw = (l31 >> 1) & 7;
  // operand fragment of the 64-deep step kk (0, 1) of a stage.  Measured on the hardware (tools/debug_mx.py): of a lane's 32 operand bytes the FIRST 16
  // belong to the instruction's first scale block and the SECOND 16 to its second one, each block being completed by the other half-wave (lane ^ 32),
  // and block b takes its scale from the lanes with (lane >> 5) == b.  So lane (row, hi) loads bytes [hi*16, +16) of MX block 2 kk and of MX block
  // 2 kk + 1 of its row (chunks kk*4 + hi and kk*4 + 2 + hi) and supplies the scale byte of block 2 kk + hi.
  int frag_off[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) frag_off[kk][h] = l31 * 128 + (((kk * 4 + h * 2 + hi) ^ sw) << 4);
  const int a_base = wm * 128 * 128, b_base = MXBM * MXBK + wn * 64 * 128;

  uint32_t a_src[4], b_src[4];                                    // byte offsets from p.A / p.W
  auto set_tile = [&](uint32_t t, int64_t& m0, int& n0) {
    const uint32_t c = min(t / chunk_tiles, n_chunks - 1), r = t - c * chunk_tiles;
    const uint32_t gw = (c == n_chunks - 1) ? p.tiles_n - c * gchunk : gchunk;
    const uint32_t tm = mp0 + r / gw, tn = c * gchunk + r % gw;
    m0 = (int64_t)tm * MXBM; n0 = (int)tn * MXBN;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 8 + piece_row;
      const int gch = slot ^ ((row >> 1) & 7);
      int64_t ar = m0 + row; if (ar > p.M - 1) ar = p.M - 1;
      int br = n0 + row; if (br > p.N - 1) br = p.N - 1;
      a_src[i] = (uint32_t)(ar * p.lda + gch * 16);
      b_src[i] = (uint32_t)((int64_t)br * p.ldw + gch * 16);
    }
  };
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane(mx_lds_addr(smem) + (wave * 4) * 1024);
  auto stage = [&](int s, int kt) {
    const uint32_t l = lds_wave + s * MX_STAGE;
    mx_dma4(a_src[0] + kt * MXBK, a_src[1] + kt * MXBK, a_src[2] + kt * MXBK, a_src[3] + kt * MXBK, p.A, l);
    mx_dma4(b_src[0] + kt * MXBK, b_src[1] + kt * MXBK, b_src[2] + kt * MXBK, b_src[3] + kt * MXBK, p.W, l + MXBM * MXBK);
  };
  // scale matrices are STAGE-major: plane kt (ld bytes apart) holds one dword per row = the four E8M0 bytes of that row's 128-deep stage kt, so the
  // scales of a stage's 256 A rows (and of its 256 W rows) are ONE contiguous KiB = one LDS-DMA piece each (waves 0 and 1), next to the stage's 64
  // operand pieces.  As six dword loads per lane they were 48 more vector-memory instructions per stage - on the path that bounds the k-loop.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const uint32_t lds_sc = __builtin_amdgcn_readfirstlane(mx_lds_addr(smem) + MX_SCALE_OFF);
  auto scale_piece = [&](int s, int kt, int64_t mm, int nn) {      // planes are padded to whole 256-row tiles (host check): no clamping
    if (wave_u == 0) mx_dma1((uint32_t)lane * 16u, p.sA + (int64_t)kt * p.ldsa + mm * 4, lds_sc + s * 2048);
    else if (wave_u == 1) mx_dma1((uint32_t)lane * 16u, p.sW + (int64_t)kt * p.ldsw + (int64_t)nn * 4, lds_sc + s * 2048 + 1024);
  };
  const int sa_rd = MX_SCALE_OFF + (wm * 128 + l31) * 4, sb_rd = MX_SCALE_OFF + 1024 + (wn * 64 + l31) * 4;

  const int nk = p.K / MXBK;
  uint32_t t = li;
  if (t >= t_end) return;
  int64_t m0; int n0;
  set_tile(t, m0, n0);
  scale_piece(0, 0, m0, n0);
  stage(0, 0);
  float* slab = reinterpret_cast<float*>(smem + MX_STAGE + wave * MX_SLAB_BYTES);
  const int ecol = (lane & 15) * 4;
  const uint32_t esz = OUT == 2 ? 1u : (OUT_BF16 ? 2u : 4u);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)(uint32_t)(p.M * p.ldc * esz), 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), (short)0, HAS_RES ? (int)(uint32_t)(p.M * p.ldr * 4) : 0, 0x00020000);
  const uint32_t cstep = (uint32_t)(4 * p.ldc) * esz, rstep = (uint32_t)(4 * p.ldr) * 4u;
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.sC, (short)0, OUT == 2 ? (int)(uint32_t)((p.N / MXBK) * p.ldsc) : 0, 0x00020000);

  for (;;) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // One 128-deep stage.  REFILL is a compile-time flag: the stage loop proper (kt < nk - 1) always refills the other slot and loads the next
    // stage's scales, the last stage is peeled - with a run-time `if` around the LDS-DMA statements hipcc splits the body into basic blocks
    // and sinks the MFMAs of the first 64-deep step below the fragment reads of the second (96 fragment registers live, ~60 spills in the loop).
    auto kstep = [&](int kt, auto refill_tag) {
      constexpr bool REFILL = decltype(refill_tag)::value;
      mx_wait_vmcnt0_barrier();                                    // stage kt (and its scale dwords) landed; slot (kt+1)&1 is free
      uint32_t sa_c[4], sb_c[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) sa_c[i] = *reinterpret_cast<const uint32_t*>(smem + sa_rd + (kt & 1) * 2048 + i * 128);
#pragma unroll
      for (int j = 0; j < 2; ++j) sb_c[j] = *reinterpret_cast<const uint32_t*>(smem + sb_rd + (kt & 1) * 2048 + j * 128);
      if (REFILL && !(SF_MX_ABL & 2)) scale_piece((kt + 1) & 1, kt + 1, m0, n0);
      const char* sa = smem + (kt & 1) * MX_STAGE + a_base;
      const char* sb = smem + (kt & 1) * MX_STAGE + b_base;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        mx_i32x8 b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const mx_i32x4 lo = *reinterpret_cast<const mx_i32x4*>(sb + j * 32 * 128 + frag_off[kk][0]);
          const mx_i32x4 hi4 = *reinterpret_cast<const mx_i32x4*>(sb + j * 32 * 128 + frag_off[kk][1]);
          b[j] = mx_i32x8{lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
        }
        const int sh = (kk * 2 + hi) * 8;                          // this lane's MX block within the stage's four
        int sbv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) sbv[j] = (int)((sb_c[j] >> sh) & 0xffu);
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {                           // A fragments two at a time
          mx_i32x8 a[2];
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = ih * 2 + ii;
            const mx_i32x4 lo = *reinterpret_cast<const mx_i32x4*>(sa + i * 32 * 128 + frag_off[kk][0]);
            const mx_i32x4 hi4 = *reinterpret_cast<const mx_i32x4*>(sa + i * 32 * 128 + frag_off[kk][1]);
            a[ii] = mx_i32x8{lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
          }
#pragma unroll
          for (int ii = 0; ii < 2; ++ii) {
            const int i = ih * 2 + ii;
            const int sav = (int)((sa_c[i] >> sh) & 0xffu);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (!(SF_MX_ABL & 4)) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[ii], b[j], acc[i][j], 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, sav, 0, sbv[j]);
              else if (j == 0) acc[i][0][0] += (float)(a[ii][0] + b[0][0] + b[1][7] + sav + sbv[0] + sbv[1]);   // ablation: operands stay live, no matrix work
            }
            if (SF_MX_DMA == 3 && REFILL && kk == 0 && !(SF_MX_ABL & 2)) {
              __builtin_amdgcn_sched_barrier(0);
              const uint32_t l = lds_wave + ((kt + 1) & 1) * MX_STAGE;
              const int ko = (kt + 1) * MXBK;
              if (ih == 0) mx_dma2(a_src[2 * ii] + ko, a_src[2 * ii + 1] + ko, p.A, l + ii * 2048);
              else mx_dma2(b_src[2 * ii] + ko, b_src[2 * ii + 1] + ko, p.W, l + MXBM * MXBK + ii * 2048);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (SF_MX_DMA == 2 && REFILL && kk == 0 && !(SF_MX_ABL & 2)) {
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t l = lds_wave + ((kt + 1) & 1) * MX_STAGE;
            const int ko = (kt + 1) * MXBK;
            if (ih == 0) mx_dma4(a_src[0] + ko, a_src[1] + ko, a_src[2] + ko, a_src[3] + ko, p.A, l);
            else mx_dma4(b_src[0] + ko, b_src[1] + ko, b_src[2] + ko, b_src[3] + ko, p.W, l + MXBM * MXBK);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (SF_MX_DMA == 0 && REFILL && !(SF_MX_ABL & 2)) {
          const uint32_t l = lds_wave + ((kt + 1) & 1) * MX_STAGE;
          const int ko = (kt + 1) * MXBK;
          if (kk == 0) mx_dma4(a_src[0] + ko, a_src[1] + ko, a_src[2] + ko, a_src[3] + ko, p.A, l);
          else mx_dma4(b_src[0] + ko, b_src[1] + ko, b_src[2] + ko, b_src[3] + ko, p.W, l + MXBM * MXBK);
          __builtin_amdgcn_sched_barrier(0);
        }
        // the refill's 8 LDS-DMA issues ride behind the MFMAs of the FIRST 64-deep step: the whole stage is requested by the middle of the k-step
        // (a 128-deep stage has only two steps - with the B pieces behind the second step's MFMAs, as in the bf16 kernel's four-step stage, they
        // were issued right in front of the next wait and their full latency showed in every k-step: qkv 1041 us at 208 segments).
        if (SF_MX_DMA == 1 && REFILL && kk == 0 && !(SF_MX_ABL & 2)) {
          const uint32_t l = lds_wave + ((kt + 1) & 1) * MX_STAGE;
          const int ko = (kt + 1) * MXBK;
          mx_dma4(a_src[0] + ko, a_src[1] + ko, a_src[2] + ko, a_src[3] + ko, p.A, l);
          mx_dma4(b_src[0] + ko, b_src[1] + ko, b_src[2] + ko, b_src[3] + ko, p.W, l + MXBM * MXBK);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    for (int kt = 0; kt + 1 < nk; ++kt) kstep(kt, std::true_type{});
    kstep(nk - 1, std::false_type{});
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int64_t em0 = m0; const int en0 = n0;
    const uint32_t tnext = t + per_xcd_blocks;
    const bool more = tnext < t_end;
    if (more) {
      set_tile(tnext, m0, n0);
      scale_piece(0, 0, m0, n0);
      stage(0, 0);
    }

    if (en0 + wn * 64 < p.N) {
      const int gcol = en0 + wn * 64 + ecol;
      const int64_t row0 = em0 + wm * 128 + (lane >> 4);
      const uint32_t coff0 = (uint32_t)(row0 * p.ldc + gcol) * esz, roff0 = (uint32_t)(row0 * p.ldr + gcol) * 4u;
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bias4 = *reinterpret_cast<const float4*>(p.bias + gcol);
      float4 res[2][4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) res[0][ps] = res[1][ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS_RES) mx_load_res<4>(res[0], rr, roff0, rstep);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int i = g >> 1, q2 = g & 1;
        if (HAS_RES && g + 1 < 8) mx_load_res<4>(res[(g + 1) & 1], rr, roff0 + (g + 1) * 4 * rstep, rstep);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int qq = 0; qq < 2; ++qq)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              slab[(qq * 8 + hi * 4 + r) * MX_EPI_LD + j * 32 + l31] = acc[i][j][(q2 * 2 + qq) * 4 + r];
        float4 v[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) v[ps] = *reinterpret_cast<const float4*>(slab + (ps * 4 + (lane >> 4)) * MX_EPI_LD + ecol);
        if (OUT == 2) {
          const int colblock = gcol >> 5;                          // this lane's 32-column block of the output row
          const int64_t left = p.M - (row0 + g * 16);            // rows_left <= 0 for every lane that must not write a scale byte
          const uint32_t sc_off = (uint32_t)((int64_t)(colblock >> 2) * p.ldsc + (colblock & 3) + (row0 + g * 16) * 4);
          mx_epi_store<OUT, GELU, HAS_RES>(v, bias4, res[g & 1], rc, coff0 + g * 4 * cstep, cstep, rsc, sc_off,
                                           (lane & 7) == 0 ? (int)(left > 64 ? 64 : left) : 0);
        } else {
          mx_epi_store<OUT, GELU, HAS_RES>(v, bias4, res[g & 1], rc, coff0 + g * 4 * cstep, cstep);
        }
      }
    }
    if (!more) break;
    t = tnext;
  }
}


// =============================================================================================================================================
// Round 3: the quadrant-phased schedule of sf_gemm_pp.hip on MXFP8 operands (same BYTE geometry: a 128-deep fp8 k-tile = 4 half-tiles of 16 KiB,
// a quadrant = 64 x 32 outputs = 4 MFMAs of 32x32x64 = the 256 matrix cycles of 8 bf16 MFMAs).  Differences to the bf16 kernel:
//   * the scale dwords of a k-tile (one per row: the four E8M0 bytes of its 128-deep stage) are a NINTH piece per wave and k-tile - 64 rows x 4 bytes
//     by global_load_lds_dword, waves 0-3 the A rows, waves 4-7 the W rows - issued with half-tile A0, all six dwords a lane needs read in phase 0
//     (so the scale buffer is free for its refill two phases later, like A0): every counted wait is vmcnt(9);
//   * a lane's scale byte of MFMA step kk is byte 2 kk + (lane >> 5) of its dword: the dwords are shifted by (lane >> 5) * 8 once per k-tile and the
//     bytes picked with constant v_bfe in the read segment (the matrix segment stays pure MFMA);
//   * LDS: 128 KiB ring + 4 KiB scale buffers + 8 x 2 KiB epilogue slabs (the epilogue walks 16 groups of 8 rows instead of 8 of 16) = 148 KiB: the
//     slabs no longer overlay an operand slot, so the load stream runs across tiles as in the bf16 kernel.
// Same products in the same order as the round-2 kernel below: bit-identical outputs (tests/test_kernels_gpu.py::test_gemm_mxfp8_schedules_bitwise).
// =============================================================================================================================================
#define MQ_HALF (128 * 128)
#define MQ_STAGE (4 * MQ_HALF)
#define MQ_SCALE_OFF (2 * MQ_STAGE)
#define MQ_SLAB_OFF (MQ_SCALE_OFF + 4096)
#define MQ_SLAB_BYTES 2048
#define MQ_LDS (MQ_SLAB_OFF + 8 * MQ_SLAB_BYTES)   // 151,552 B
#ifndef SF_MX_STORECNT
#define SF_MX_STORECNT 1
#endif

__device__ __forceinline__ void mq_dma2(uint32_t v0, uint32_t v1, const void* sbase, uint32_t l0) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %4\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(v0), "v"(v1), "s"(sbase), "s"(l0) : "memory", "scc");
}
// 64 lanes x 4 bytes (a 256-byte piece): lane offset in %1, SGPR base
__device__ __forceinline__ void mq_dma_dword(uint32_t v0, const void* sbase, uint32_t l0) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "s"(sbase), "s"(l0) : "memory");
}
template <int N>
__device__ __forceinline__ void mq_wait_vmcnt() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}
__device__ __forceinline__ void mq_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int V> using mq_ic = std::integral_constant<int, V>;

template <int OUT, bool GELU, bool HAS_RES>
__global__ __launch_bounds__(512, 2) void gemm_mxfp8_pp_kernel(MxArgs p) {
  constexpr bool OUT_BF16 = OUT == 1;
  // WIDEMX (MXFP8 output without residual - fc1 + GELU): the accumulator blocks are computed TRANSPOSED (W fragment as the A operand: lanes = tokens, a lane's 16
  // registers = 16 of the 32 features of one scale block, the lane 32 away holds the other 16), so that the quantisation needs one cross-lane step per block and no
  // fp32 staging (the fp8 bytes alone pass through the slab, 2 KiB per 32 tokens): 8 row stores of 16 bytes + 4 scale stores of 2 bytes per wave instead of 32 + 32 (was 1370 us for fc1 of the 13-segment batch against 677 for qkv)
  constexpr bool WIDEMX = OUT == 2 && !HAS_RES;
  // WIDEBF (bf16 output without residual - the spatial qkv projection): the same transposed blocks, bf16 pairs through the 2-KiB slab, 16-byte row stores (16 per wave
  // instead of 32 stores of 8 bytes behind an fp32 slab round trip); bit-identical to the general epilogue
  constexpr bool WIDEBF = OUT == 1 && !HAS_RES;
  constexpr bool TRANSPOSED = WIDEMX || WIDEBF;
  constexpr int EPI_VM = WIDEMX ? 12 : (WIDEBF ? 16 : (OUT == 2 ? 54 : 32));     // vector-memory operations of one epilogue (OUT 2 with residual: 64, capped by the 6-bit counter: 9 + 54 = 63)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l31 = lane & 31, hi = lane >> 5;

  const uint32_t xcd = blockIdx.x & 7u, li = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
  const uint32_t tiles_m = p.tiles_total / p.tiles_n;
  const uint32_t mp8 = (tiles_m + 7u) >> 3;
  const uint32_t mp0 = min(xcd * mp8, tiles_m), mp1 = min(mp0 + mp8, tiles_m), n_mp = mp1 - mp0;
  const uint32_t gchunk = p.nchunk ? min(p.nchunk, p.tiles_n) : p.tiles_n;
  const uint32_t n_chunks = (p.tiles_n + gchunk - 1) / gchunk, chunk_tiles = n_mp * gchunk;
  const uint32_t t_end = n_mp * p.tiles_n;
  auto tile_origin = [&](uint32_t t, int64_t& m0, int& n0) {
    const uint32_t c = min(t / chunk_tiles, n_chunks - 1), r = t - c * chunk_tiles;
    const uint32_t gw = (c == n_chunks - 1) ? p.tiles_n - c * gchunk : gchunk;
    const uint32_t tm = mp0 + r / gw, tn = c * gchunk + r % gw;
    m0 = (int64_t)tm * 256; n0 = (int)tn * 256;
  };

  // fragment offsets inside the current stage (flipped by bit 16 once per k-tile): [kk][h] = chunk kk * 4 + h * 2 + hi of the lane's row
  const int sw = (l31 >> 1) & 7;
  int a_off[2][2], b_off[2][2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      a_off[kk][h] = (wm * 64 + l31) * 128 + (((kk * 4 + h * 2 + hi) ^ sw) << 4);                  // + ha * MQ_HALF + i * 4096
      b_off[kk][h] = 2 * MQ_HALF + (wn * 32 + l31) * 128 + (((kk * 4 + h * 2 + hi) ^ sw) << 4);     // + hb * MQ_HALF
    }

  // ---- load iterator ----
  const int nk = p.K / 128;
  uint32_t ld_t = li;
  bool ld_ok = ld_t < t_end;
  if (!ld_ok) return;
  int ld_kt = 0;
  const char* ldA = nullptr; const char* ldW = nullptr; const char* ldS = nullptr;   // ldS: this wave's 64 scale dwords of the tile (A rows or W rows)
  uint32_t oA[2][2], oW[2][2];
  const int64_t ld_sstep = wave < 4 ? p.ldsa : p.ldsw;
  auto ld_set = [&](uint32_t t) {
    int64_t m0; int n0;
    tile_origin(t, m0, n0);
    ldA = reinterpret_cast<const char*>(p.A) + m0 * p.lda;
    ldW = reinterpret_cast<const char*>(p.W) + (int64_t)n0 * p.ldw;
    ldS = wave < 4 ? reinterpret_cast<const char*>(p.sA) + (m0 + wave * 64) * 4 : reinterpret_cast<const char*>(p.sW) + ((int64_t)n0 + (wave - 4) * 64) * 4;
    const int64_t mleft = p.M - 1 - m0;
    const int mrem = mleft < 255 ? (int)mleft : 255, nrem = min(p.N - 1 - n0, 255);
    int ltid = threadIdx.x;
    asm volatile("" : "+v"(ltid));
    const int lr = (ltid & 63) >> 3, lc = ltid & 7;
    const uint32_t lda1 = (uint32_t)p.lda, ldw1 = (uint32_t)p.ldw;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wave * 16 + j * 8 + lr;
      const uint32_t gch = (uint32_t)((lc ^ ((r >> 1) & 7)) << 4);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int tr = min((r >> 6) * 128 + h * 64 + (r & 63), mrem);
        const int tc = min((r >> 5) * 64 + h * 32 + (r & 31), nrem);
        oA[h][j] = __umul24((uint32_t)tr, lda1) + gch;
        oW[h][j] = __umul24((uint32_t)tc, ldw1) + gch;
      }
    }
  };
  const uint32_t lds0 = __builtin_amdgcn_readfirstlane(mx_lds_addr(smem));
  const uint32_t lds_wave = lds0 + wave * 2048;
  const uint32_t lds_sc_w = lds0 + MQ_SCALE_OFF + wave * 256;      // + stage * 2048: [A rows 0-255 | W rows 0-255] dwords
  const uint32_t lane4 = (uint32_t)lane * 4u;
  // PART: 0 = A0 (+ the k-tile's scale piece), 1 = B0, 2 = B1, 3 = A1
  // Branch-free (as sf_gemm_pp.hip's SLIM): the k-tile bases run along (curA / curW / curS); a DRY iterator (last k-tiles of the workgroup's last tile) keeps
  // issuing - the same pieces of the last valid k-tile into the ring buffers that are free for a refill by construction - so that every counted wait keeps
  // its count and no read segment carries an `issued` branch; vmcnt(0) in front of that tile's epilogue retires them before the workgroup can end
  const char* curA = nullptr; const char* curW = nullptr; const char* curS = nullptr;
  auto issue = [&](auto PARTc, auto STc) -> bool {
    constexpr int PART = decltype(PARTc)::value, ST = decltype(STc)::value;
    constexpr bool isA = PART == 0 || PART == 3;
    constexpr int h = PART >= 2 ? 1 : 0;
    const uint32_t l = lds_wave + ST * MQ_STAGE + (isA ? h : 2 + h) * MQ_HALF;
    if (isA) mq_dma2(oA[h][0], oA[h][1], curA, l);
    else mq_dma2(oW[h][0], oW[h][1], curW, l);
    if (PART == 0) mq_dma_dword(lane4, curS, lds_sc_w + ST * 2048);
    if (PART == 3 && ld_ok) {
      if (++ld_kt == nk) {
        ld_kt = 0;
        ld_t += per_xcd_blocks;
        ld_ok = ld_t < t_end;
        if (ld_ok) { ld_set(ld_t); curA = ldA; curW = ldW; curS = ldS; }
      } else {
        curA += 128; curW += 128; curS += ld_sstep;
      }
    }
    return true;
  };

  uint32_t t = li;
  int64_t m0; int n0;
  tile_origin(t, m0, n0);
  char* bslab = smem + MQ_SLAB_OFF + wave * MQ_SLAB_BYTES;
  const uint32_t slab_lds = __builtin_amdgcn_readfirstlane(mx_lds_addr(bslab));
  const uint32_t esz = OUT == 2 ? 1u : (OUT_BF16 ? 2u : 4u);
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(p.C, (short)0, (int)(uint32_t)(p.M * p.ldc * esz), 0x00020000);
  const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.R), (short)0, HAS_RES ? (int)(uint32_t)(p.M * p.ldr * 4) : 0, 0x00020000);
  const uint32_t cstep = (uint32_t)(4 * p.ldc) * esz, rstep = (uint32_t)(4 * p.ldr) * 4u;
  const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc(p.sC, (short)0, OUT == 2 ? (int)(uint32_t)((p.N / MXBK) * p.ldsc) : 0, 0x00020000);
  const bool has_bias = p.bias != nullptr;
  // the wave's 64 bias values: one 256-byte LDS-DMA piece into the (idle) slab at the top of the tile - no compiler-visible load, no compiler vmcnt(0)
  auto issue_bias = [&](int n0_) {
    if (has_bias) mq_dma_dword(lane4, reinterpret_cast<const char*>(p.bias + min(n0_ + wn * 64, p.N - 64)), slab_lds);
  };

  ld_set(ld_t);
  curA = ldA; curW = ldW; curS = ldS;
  issue_bias(n0);
  issue(mq_ic<0>{}, mq_ic<0>{}); issue(mq_ic<1>{}, mq_ic<0>{}); issue(mq_ic<2>{}, mq_ic<0>{}); issue(mq_ic<3>{}, mq_ic<0>{});
  issue(mq_ic<0>{}, mq_ic<1>{}); issue(mq_ic<1>{}, mq_ic<1>{});
  mq_wait_vmcnt<9>();                                             // A0 (+ scales) | B0 of k-tile 0 have landed: B1, A1, A0 + scales, B0 (2 + 2 + 3 + 2) are younger
  mq_barrier();
  int extra = 0;

  for (;;) {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // (opaque to the optimiser: folding the zeros into the first MFMAs' C operands makes hipcc peel the first k-tile pair - a second copy of the loop
    // body, and 165-323 spilled registers)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" : "+v"(acc[i][j]));
    mx_i32x8 a[2][2], b0[2], b1[2];                               // [i][kk], [kk]
    uint32_t sa[2][2], sb[2];                                     // scale dwords [ha][i], [hb], shifted so that byte 2 kk is this lane's block of step kk

    auto wait_loads = [&](bool issued, bool first) {
      if (!issued) mq_wait_vmcnt<0>();
      else if (SF_MX_STORECNT && first && extra) mq_wait_vmcnt<9 + EPI_VM>();
      else mq_wait_vmcnt<9>();
    };
    auto frag = [&](const char* base, const int (&off)[2][2], int kk) -> mx_i32x8 {
      const mx_i32x4 lo = *reinterpret_cast<const mx_i32x4*>(base + off[kk][0]);
      const mx_i32x4 hi4 = *reinterpret_cast<const mx_i32x4*>(base + off[kk][1]);
      return mx_i32x8{lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
    };
    auto mma = [&](auto HAc, auto HBc, const mx_i32x8 (&bf)[2], const int (&sav)[2][2], const int (&sbv)[2]) {
      constexpr int HA = decltype(HAc)::value, HB = decltype(HBc)::value;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          acc[HA * 2 + i][HB] = TRANSPOSED ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bf[kk], a[i][kk], acc[HA * 2 + i][HB], 0, 0, 0, sbv[kk], 0, sav[i][kk])
                                       : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i][kk], bf[kk], acc[HA * 2 + i][HB], 0, 0, 0, sav[i][kk], 0, sbv[kk]);
      // the MFMAs are pure to the optimiser: with the run-time branches of the read segments around, LLVM sinks them towards their next use (the next
      // k-tile's MFMAs on the same accumulator) - out of the matrix segment, with every fragment live across phases; an opaque use pins them here
      asm volatile("" : "+v"(acc[HA * 2][HB]), "+v"(acc[HA * 2 + 1][HB]));
      __builtin_amdgcn_s_setprio(0);
    };
    auto pick = [&](uint32_t dw, int kk) -> int { return (int)((dw >> (kk * 16)) & 0xffu); };
    auto ktile = [&](auto Sc, bool first) {
      constexpr int S = decltype(Sc)::value;
      const char* st = smem;                                        // a_off / b_off carry the stage
      int sav[2][2], sbv[2];
      // ---- phase 0: (A0, B0); all scale dwords of the k-tile ----
      {
        const char* scp = smem + MQ_SCALE_OFF + S * 2048;
#pragma unroll
        for (int ha = 0; ha < 2; ++ha)
#pragma unroll
          for (int i = 0; i < 2; ++i) sa[ha][i] = *reinterpret_cast<const uint32_t*>(scp + (wm * 128 + ha * 64 + i * 32 + l31) * 4) >> (hi * 8);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) sb[hb] = *reinterpret_cast<const uint32_t*>(scp + 1024 + (wn * 64 + hb * 32 + l31) * 4) >> (hi * 8);
      }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) b0[kk] = frag(st, b_off, kk);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i][kk] = frag(st + i * 4096, a_off, kk);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) { sbv[kk] = pick(sb[0], kk); sav[0][kk] = pick(sa[0][0], kk); sav[1][kk] = pick(sa[0][1], kk); }
      __builtin_amdgcn_sched_barrier(0);
      wait_loads(issue(mq_ic<2>{}, mq_ic<S ^ 1>{}), first);        // B1(kt+1) issued; B1(kt) landed
      mq_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(mq_ic<0>{}, mq_ic<0>{}, b0, sav, sbv);
      __builtin_amdgcn_sched_barrier(0);
      mq_barrier();
      // ---- phase 1: (A0, B1) ----
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) b1[kk] = frag(st + MQ_HALF, b_off, kk);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) sbv[kk] = pick(sb[1], kk);
      __builtin_amdgcn_sched_barrier(0);
      wait_loads(issue(mq_ic<3>{}, mq_ic<S ^ 1>{}), first);        // A1(kt+1) issued; A1(kt) landed
      mq_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(mq_ic<0>{}, mq_ic<1>{}, b1, sav, sbv);
      __builtin_amdgcn_sched_barrier(0);
      mq_barrier();
      // ---- phase 2: (A1, B1); the fragment addresses move on to the other stage ----
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i][kk] = frag(st + MQ_HALF + i * 4096, a_off, kk);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) { sav[0][kk] = pick(sa[1][0], kk); sav[1][kk] = pick(sa[1][1], kk); }
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(a_off[kk][h]));
          asm volatile("v_xor_b32 %0, 0x10000, %0" : "+v"(b_off[kk][h]));
        }
      __builtin_amdgcn_sched_barrier(0);
      issue(mq_ic<0>{}, mq_ic<S>{});                               // A0 + scales of k-tile kt+2; phase 3 reads nothing: no wait
      mq_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(mq_ic<1>{}, mq_ic<1>{}, b1, sav, sbv);
      __builtin_amdgcn_sched_barrier(0);
      mq_barrier();
      // ---- phase 3: (A1, B0) ----
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) sbv[kk] = pick(sb[0], kk);
      wait_loads(issue(mq_ic<1>{}, mq_ic<S>{}), first);            // B0(kt+2) issued; A0 + scales | B0 of k-tile kt+1 landed
      mq_barrier();
      __builtin_amdgcn_sched_barrier(0);
      mma(mq_ic<1>{}, mq_ic<0>{}, b0, sav, sbv);
      __builtin_amdgcn_sched_barrier(0);
      mq_barrier();
    };

    if (wm == 1) mq_barrier();                                    // the wm = 1 waves run one barrier behind
    int kt_first = 0;
    asm volatile("" : "+s"(kt_first));                             // opaque zero: `kt == 0` lets hipcc peel the first iteration - a second copy of the body, with spills
    for (int kt = 0; kt < nk; kt += 2) {
      ktile(mq_ic<0>{}, kt == kt_first);
      ktile(mq_ic<1>{}, false);
    }
    if (wm == 0) mq_barrier();

    if (!ld_ok) mq_wait_vmcnt<0>();                                // a dry iterator's pieces must have landed before this workgroup's LDS can be handed on
    const int64_t em0 = m0; const int en0 = n0;
    extra = 0;
    if (WIDEBF) {
      if (en0 + wn * 64 < p.N) {
        extra = 1;
        int etid = threadIdx.x;
        asm volatile("" : "+v"(etid));
        const int el = etid & 63, el31 = el & 31, ehi = el >> 5;
        float4 bia[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            bia[j][g] = has_bias ? *reinterpret_cast<const float4*>(bslab + (j * 32 + g * 8 + ehi * 4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);   // landed long ago
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the slab is reused below
        // slab: 32 tokens x 64 bytes (32 features of block j as bf16); 16-byte chunk g (features g * 8 .. + 7: the low lane's four, then the high lane's four) of row t at
        // slot g ^ ((t >> 1) & 3)
        const int wr_off = el31 * 64 + ehi * 8, wsw = (el31 >> 1) & 3;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              sf_f32x2_t g0 = {acc[ib][j][g * 4 + 0] + bia[j][g].x, acc[ib][j][g * 4 + 1] + bia[j][g].y};
              sf_f32x2_t g1 = {acc[ib][j][g * 4 + 2] + bia[j][g].z, acc[ib][j][g * 4 + 3] + bia[j][g].w};
              if (GELU) gelu_erf4(g0, g1);
              mx_u32x2 w2; w2.x = pack_bf2(g0.x, g0.y); w2.y = pack_bf2(g1.x, g1.y);
              *reinterpret_cast<mx_u32x2*>(bslab + wr_off + ((g ^ wsw) << 4)) = w2;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int tr = h * 16 + (el >> 2);
              const mx_u32x4 o = *reinterpret_cast<const mx_u32x4*>(bslab + tr * 64 + (((el & 3) ^ ((tr >> 1) & 3)) << 4));
              __builtin_amdgcn_raw_buffer_store_b128(o, rc, (uint32_t)((em0 + wm * 128 + ib * 32 + tr) * p.ldc + en0 + wn * 64 + j * 32 + (el & 3) * 8) * 2u, 0, SF_MX_STORE_AUX);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
      }
    } else if (WIDEMX) {
      if (en0 + wn * 64 < p.N) {
        extra = 1;
        int etid = threadIdx.x;
        asm volatile("" : "+v"(etid));
        const int el = etid & 63, el31 = el & 31, ehi = el >> 5;
        float4 bia[2][4];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            bia[j][g] = has_bias ? *reinterpret_cast<const float4*>(bslab + (j * 32 + g * 8 + ehi * 4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);   // landed long ago
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the slab is reused below
        const int64_t row0 = em0 + wm * 128 + el31;
        const int colblock = (en0 + wn * 64) >> 5;                                    // even: the wave's two scale bytes of a row are adjacent
        const uint32_t sc_base = (uint32_t)((int64_t)(colblock >> 2) * p.ldsc + (colblock & 3));
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
          const int64_t row = row0 + ib * 32;
          uint32_t be2 = 0;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            float f[16];
            float amax = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              sf_f32x2_t g0 = {acc[ib][j][g * 4 + 0] + bia[j][g].x, acc[ib][j][g * 4 + 1] + bia[j][g].y};
              sf_f32x2_t g1 = {acc[ib][j][g * 4 + 2] + bia[j][g].z, acc[ib][j][g * 4 + 3] + bia[j][g].w};
              if (GELU) gelu_erf4(g0, g1);
              const uint32_t p01 = pack_bf2(g0.x, g0.y), p23 = pack_bf2(g1.x, g1.y);   // quantised from the bf16-rounded value, as sf_quantize_mxfp8 would
              f[g * 4 + 0] = __uint_as_float(p01 << 16); f[g * 4 + 1] = __uint_as_float(p01 & 0xffff0000u);
              f[g * 4 + 2] = __uint_as_float(p23 << 16); f[g * 4 + 3] = __uint_as_float(p23 & 0xffff0000u);
              amax = fmaxf(fmaxf(amax, fmaxf(fabsf(f[g * 4 + 0]), fabsf(f[g * 4 + 1]))), fmaxf(fabsf(f[g * 4 + 2]), fabsf(f[g * 4 + 3])));
            }
            {                                                                          // the block's other 16 values live in the lane 32 away
              const auto sw2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(amax), __float_as_uint(amax), false, false);
              amax = fmaxf(__uint_as_float(sw2[0]), __uint_as_float(sw2[1]));
            }
            int be = sf_mx_be(amax);
            be = be < 1 ? 1 : (be > 254 ? 254 : be);
            const float inv = __uint_as_float((uint32_t)(254 - be) << 23);
            be2 |= (uint32_t)be << (8 * j);
            uint32_t d[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              int w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f[g * 4 + 0] * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(f[g * 4 + 1] * inv, 448.f, -448.f), 0, false);
              w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(f[g * 4 + 2] * inv, 448.f, -448.f), __builtin_amdgcn_fmed3f(f[g * 4 + 3] * inv, 448.f, -448.f), w, true);
              d[g] = (uint32_t)w;
            }
            // lanes 0-31 hold features g * 8 + 0..3, lanes 32-63 g * 8 + 4..7: permlane32_swap(x, z) = {x.lo | z.lo, x.hi | z.hi} puts features 0..15 of the block into
            // the low lane and 16..31 into the high lane, in order
            const auto s02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            const auto s13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            mx_u32x4 o; o.x = s02[0]; o.y = s02[1]; o.z = s13[0]; o.w = s13[1];
            // through the wave's 2-KiB slab (32 tokens x 64 bytes, 16-byte chunk c of row t at slot c ^ ((t >> 1) & 3)): stored from the registers a row would go
            // out as 32-byte pieces, 32 write requests per instruction (measured: slower than the 4-byte stores it replaced); from the slab a store is 16 rows x 64 bytes
            *reinterpret_cast<mx_u32x4*>(bslab + el31 * 64 + (((j * 2 + ehi) ^ ((el31 >> 1) & 3)) << 4)) = o;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int tr = h * 16 + (el >> 2);
            const mx_u32x4 o = *reinterpret_cast<const mx_u32x4*>(bslab + tr * 64 + (((el & 3) ^ ((tr >> 1) & 3)) << 4));
            __builtin_amdgcn_raw_buffer_store_b128(o, rc, (uint32_t)((em0 + wm * 128 + ib * 32 + tr) * p.ldc + en0 + wn * 64 + (el & 3) * 16), 0, SF_MX_STORE_AUX);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          // the row's two scale bytes (blocks j = 0, 1) from the low lane; the high lane and rows >= M (they would land in the next plane) get an offset the range check drops
          __builtin_amdgcn_raw_buffer_store_b16((unsigned short)be2, rsc, (ehi == 0 && row < p.M) ? sc_base + (uint32_t)row * 4u : 0xffffffffu, 0, 0);
        }
      }
    } else if (en0 + wn * 64 < p.N) {
      extra = 1;
      int etid = threadIdx.x;
      asm volatile("" : "+v"(etid));
      const int el = etid & 63, el31 = el & 31, ehi = el >> 5, ecol = (el & 15) * 4;
      float* slab = reinterpret_cast<float*>(bslab);
      const int gcol = en0 + wn * 64 + ecol;
      const int64_t row0 = em0 + wm * 128 + (el >> 4);
      const uint32_t coff0 = (uint32_t)(row0 * p.ldc + gcol) * esz, roff0 = (uint32_t)(row0 * p.ldr + gcol) * 4u;
      float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (has_bias) bias4 = *reinterpret_cast<const float4*>(slab + ecol);      // landed long ago: every counted wait of the k-loop covered it
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      float4 res[2][2];
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) res[0][ps] = res[1][ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (HAS_RES) mx_load_res<2>(res[0], rr, roff0, rstep);
#pragma unroll
      for (int g = 0; g < 16; ++g) {                               // rows g * 8 .. + 7 of the wave's 128
        const int i = g >> 2, q4 = g & 3;
        if (HAS_RES && g + 1 < 16) mx_load_res<2>(res[(g + 1) & 1], rr, roff0 + (g + 1) * 2 * rstep, rstep);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[(ehi * 4 + r) * MX_EPI_LD + j * 32 + el31] = acc[i][j][q4 * 4 + r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float4 v[2];
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) v[ps] = *reinterpret_cast<const float4*>(slab + (ps * 4 + (el >> 4)) * MX_EPI_LD + ecol);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (OUT == 2) {
          const int colblock = gcol >> 5;
          const int64_t left = p.M - (row0 + g * 8);
          const uint32_t sc_off = (uint32_t)((int64_t)(colblock >> 2) * p.ldsc + (colblock & 3) + (row0 + g * 8) * 4);
          mx_epi_store<OUT, GELU, HAS_RES, 2>(v, bias4, res[g & 1], rc, coff0 + g * 2 * cstep, cstep, rsc, sc_off, (el & 7) == 0 ? (int)(left > 64 ? 64 : left) : 0);
        } else {
          mx_epi_store<OUT, GELU, HAS_RES, 2>(v, bias4, res[g & 1], rc, coff0 + g * 2 * cstep, cstep);
        }
      }
    }
    t += per_xcd_blocks;
    if (t >= t_end) break;
    tile_origin(t, m0, n0);
    issue_bias(n0);
  }
}

static thread_local int g_mx_force_sched = -1;   // test hook (per calling thread): -1 default, 0 round 2's loop, 1 quadrant-phased
extern "C" void sf_gemm_mx_force_schedule(int sched) { g_mx_force_sched = sched; }

template <int OUT, bool GELU, bool HAS_RES>
static int mx_launch(MxArgs a, hipStream_t s) {
  auto kern = gemm_mxfp8_persistent_kernel<OUT, GELU, HAS_RES>;
  auto kern_pp = gemm_mxfp8_pp_kernel<OUT, GELU, HAS_RES>;
  if (int rc = sf_prepare_kernel((const void*)kern, MX_LDS, "sf_gemm_mxfp8")) return rc;
  if (int rc = sf_prepare_kernel((const void*)kern_pp, MQ_LDS, "sf_gemm_mxfp8")) return rc;
  const int n_cu = sf_cu_count("sf_gemm_mxfp8");
  if (n_cu <= 0) return -1;
  const int64_t tiles_m = (a.M + MXBM - 1) / MXBM;
  a.tiles_n = (uint32_t)((a.N + MXBN - 1) / MXBN);
  const int64_t total = tiles_m * a.tiles_n;
  if (total >= ((int64_t)1 << 31)) { sf_set_error("sf_gemm_mxfp8: too many tiles"); return -1; }
  a.tiles_total = (uint32_t)total;
  // column-chunked sweeps as in the bf16 kernel: keep the weight slice of a sweep (nchunk * 256 rows * K bytes) within ~2.4 MB of the XCD's L2
  a.nchunk = a.K <= 2048 ? (uint32_t)(2400000 / (256 * a.K) > 0 ? 2400000 / (256 * a.K) : 1) : 0u;
  int64_t blocks = (n_cu / 8) * 8;
  if (blocks < 8) blocks = 8;                                    // (a device / partition with fewer than 8 CUs: never an empty grid)
  const int64_t need = ((total + 7) / 8) * 8;
  if (blocks > need) blocks = need;
  static int env_sched = -1;
  if (env_sched < 0) { const char* e = getenv("SF_MX_SCHED"); env_sched = e ? atoi(e) : 1; }
  // quadrant-phased: whole pairs of 128-deep k-tiles (static stage indices), 24-bit row offsets
  const bool pp = (g_mx_force_sched >= 0 ? g_mx_force_sched != 0 : env_sched != 0) && (a.K % 256) == 0 && a.lda < (1 << 23) && a.ldw < (1 << 23);
  if (pp) hipLaunchKernelGGL(kern_pp, dim3((unsigned)blocks), dim3(512), MQ_LDS, s, a);
  else hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), MX_LDS, s, a);
  SF_LAUNCH_CHECK();
  return 0;
}

extern "C" int sf_gemm_mxfp8(const uint8_t* A, int64_t lda, const uint8_t* sA, int64_t ldsa, const uint8_t* W, int64_t ldw, const uint8_t* sW,
                             int64_t ldsw, const float* bias, void* C, int c_dtype, int64_t ldc, uint8_t* sC, int64_t ldsc, const float* R, int64_t ldr,
                             int epilogue, int64_t M, int64_t N, int64_t K, void* stream) {
  SF_CHECK_ARG(A && sA && W && sW && C, "sf_gemm_mxfp8: null pointer");
  SF_CHECK_ARG(c_dtype == SF_BF16 || c_dtype == SF_F32 || c_dtype == SF_U8, "sf_gemm_mxfp8: c_dtype must be bf16, f32 or u8 (MXFP8 output)");
  SF_CHECK_ARG(c_dtype != SF_U8 || (sC && !R && (N % 128) == 0 && (ldsc % 4) == 0 && ldsc >= M * 4),
               "sf_gemm_mxfp8: MXFP8 output needs scale planes (>= M * 4 bytes each), N %% 128 == 0 and no residual");
  SF_CHECK_ARG(epilogue == SF_EPI_NONE || epilogue == SF_EPI_GELU, "sf_gemm_mxfp8: bad epilogue %d", epilogue);
  SF_CHECK_ARG(K > 0 && (K % MXBK) == 0, "sf_gemm_mxfp8: K=%lld must be a positive multiple of 128", (long long)K);
  SF_CHECK_ARG((lda % 16) == 0 && (ldw % 16) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0,
               "sf_gemm_mxfp8: A / W rows must be 16-byte aligned");
  SF_CHECK_ARG((ldsa % 16) == 0 && (ldsw % 16) == 0 && ldsa >= ((M + 255) / 256) * 1024 && ldsw >= ((N + 255) / 256) * 1024 && ((uintptr_t)sA % 16) == 0 &&
                   ((uintptr_t)sW % 16) == 0,
               "sf_gemm_mxfp8: scale planes must hold 4 bytes per row with the rows PADDED to whole 256-row tiles (plane stride >= ceil(rows / 256) * 1024 "
               "bytes, 16-byte aligned): a tile's scales are fetched as one KiB");
  SF_CHECK_ARG((N % 64) == 0 && (ldc % 4) == 0 && (!R || (ldr % 4) == 0) && ((uintptr_t)C % 16) == 0 && (!R || ((uintptr_t)R % 16) == 0) &&
                   (!bias || ((uintptr_t)bias % 16) == 0),
               "sf_gemm_mxfp8: N %% 64 == 0 and 16-byte aligned outputs are required");
  if (M <= 0 || N <= 0) return 0;
  const int64_t m_pad = ((M + 255) / 256) * 256;
  SF_CHECK_ARG(m_pad * ldc * (c_dtype == SF_U8 ? 1 : c_dtype == SF_BF16 ? 2 : 4) < ((int64_t)1 << 32) && (!R || m_pad * ldr * 4 < ((int64_t)1 << 32)),
               "sf_gemm_mxfp8: C / R must stay below 4 GiB");
  SF_CHECK_ARG(c_dtype != SF_U8 || (N / MXBK) * ldsc < ((int64_t)1 << 31), "sf_gemm_mxfp8: the output scale planes must stay below 2 GiB");
  SF_CHECK_ARG(M * lda < ((int64_t)1 << 32) && N * ldw < ((int64_t)1 << 32) && (K / MXBK) * ldsa < ((int64_t)1 << 31) && (K / MXBK) * ldsw < ((int64_t)1 << 31),
               "sf_gemm_mxfp8: operands and scale matrices must stay below 4 GiB (32-bit lane offsets)");
  MxArgs a;
  a.A = A; a.lda = lda; a.sA = sA; a.ldsa = ldsa; a.W = W; a.ldw = ldw; a.sW = sW; a.ldsw = ldsw; a.bias = bias; a.C = C; a.ldc = ldc;
  a.sC = sC; a.ldsc = ldsc; a.R = R; a.ldr = ldr; a.M = M; a.N = (int)N; a.K = (int)K; a.tiles_n = a.tiles_total = a.nchunk = 0;
  hipStream_t s = (hipStream_t)stream;
  const bool gelu = epilogue == SF_EPI_GELU, res = R != nullptr, obf = c_dtype == SF_BF16;
  if (c_dtype == SF_U8) return gelu ? mx_launch<2, true, false>(a, s) : mx_launch<2, false, false>(a, s);
  if (obf) {
    if (gelu) return res ? mx_launch<1, true, true>(a, s) : mx_launch<1, true, false>(a, s);
    return res ? mx_launch<1, false, true>(a, s) : mx_launch<1, false, false>(a, s);
  }
  if (gelu) return res ? mx_launch<0, true, true>(a, s) : mx_launch<0, true, false>(a, s);
  return res ? mx_launch<0, false, true>(a, s) : mx_launch<0, false, false>(a, s);
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Quantiser: bf16 (rows x K, K % 128 == 0) -> OCP e4m3 elements + one E8M0 scale byte per 32 consecutive k (OCP Microscaling MXFP8):
//   e = floor(log2(max |x| over the block)) - 8   (8 = exponent of the largest e4m3 normal, 448 = 1.75 * 2^8),  scale byte = e + 127,
//   q = round-to-nearest-even( x * 2^-e ) saturated to +-448 (v_cvt_pk_fp8_f32, OCP on gfx950).  An all-zero block gets scale 2^-126.
// Scales are written STAGE-major for the GEMM: scales[(k / 128) * lds + row * 4 + (k / 32) % 4].  One lane per block (64 B in, 32 B out); the
// four lanes of a 128-deep stage combine their bytes and store one dword.
// ------------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void quantize_mxfp8_kernel(const bf16_t* __restrict__ x, int64_t ldx, uint8_t* __restrict__ q, int64_t ldq,
                                                              uint8_t* __restrict__ sc, int64_t lds, int64_t rows, int kblocks) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < rows * kblocks;
  if (!live) i = rows * kblocks - 1;                               // keep the quad complete for the shuffle below
  const int64_t r = i / kblocks;
  const int kb = (int)(i - r * kblocks);
  const uint4* src = reinterpret_cast<const uint4*>(x + r * ldx + kb * 32);
  float v[32];
  float amax = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const uint4 t = src[c];
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[c * 8 + e * 2] = __uint_as_float(w[e] << 16);
      v[c * 8 + e * 2 + 1] = __uint_as_float(w[e] & 0xffff0000u);
    }
  }
#pragma unroll
  for (int e = 0; e < 32; ++e) amax = fmaxf(amax, fabsf(v[e]));
  int be = sf_mx_be(amax);      // biased exponent of amax, minus emax(e4m3); subnormal / zero amax -> clamp
  if (be < 1) be = 1;
  if (be > 254) be = 254;
  const float inv = __uint_as_float((uint32_t)(254 - be) << 23);  // 2^-(be - 127)
  uint32_t out[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    // the block maximum lands in [256, 512): saturate to the largest e4m3 normal first (the conversion itself would produce NaN above 448)
    const float f0 = __builtin_amdgcn_fmed3f(v[e * 4] * inv, 448.f, -448.f), f1 = __builtin_amdgcn_fmed3f(v[e * 4 + 1] * inv, 448.f, -448.f);
    const float f2 = __builtin_amdgcn_fmed3f(v[e * 4 + 2] * inv, 448.f, -448.f), f3 = __builtin_amdgcn_fmed3f(v[e * 4 + 3] * inv, 448.f, -448.f);
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f2, f3, w, true);
    out[e] = (uint32_t)w;
  }
  if (live) {
    uint4* dst = reinterpret_cast<uint4*>(q + r * ldq + kb * 32);
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
  }
  // kblocks % 4 == 0 and the block index is the fastest thread dimension: lanes 4j .. 4j+3 hold the four blocks of one (row, stage)
  uint32_t word = (uint32_t)be << ((kb & 3) * 8);
  word |= __shfl_xor(word, 1, 64);
  word |= __shfl_xor(word, 2, 64);
  if (live && (kb & 3) == 0) *reinterpret_cast<uint32_t*>(sc + (int64_t)(kb >> 2) * lds + r * 4) = word;
}

extern "C" int sf_quantize_mxfp8(const bf16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, uint8_t* scales, int64_t lds, int64_t rows, int64_t K,
                                 void* stream) {
  SF_CHECK_ARG(x && q && scales, "sf_quantize_mxfp8: null pointer");
  SF_CHECK_ARG(K > 0 && (K % 128) == 0 && (ldx % 8) == 0 && (ldq % 16) == 0 && (lds % 4) == 0 && lds >= rows * 4 && ((uintptr_t)x % 16) == 0 &&
                   ((uintptr_t)q % 16) == 0 && ((uintptr_t)scales % 4) == 0,
               "sf_quantize_mxfp8: K %% 128 == 0, 16-byte aligned rows and scale planes of >= rows * 4 bytes are required");
  if (rows <= 0) return 0;
  const int64_t n = rows * (K / 32);
  hipLaunchKernelGGL(quantize_mxfp8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, q, ldq, scales, lds, rows,
                     (int)(K / 32));
  SF_LAUNCH_CHECK();
  return 0;
}
